/*
 * rt_trace_kernels.h -- device code of the hot path: the fused trace kernel
 * (host-seeded, device-generated and compacting forms) and the kernels that
 * put launch rays into row 0.  Included by rt_engine.hip; the per-ray
 * arithmetic lives in rt_math.h.
 */
#ifndef RT_TRACE_KERNELS_H
#define RT_TRACE_KERNELS_H

#include "rt_march.h"

/* the rows are written once and not read again by the kernel: non-temporal
 * stores.  In mixed memory (rt_place.h) 1.100 -> 1.025 ms for the headline
 * trace, same process, two builds; in one-class memory they had been worth
 * +-0.3 % (round 2).  -DRT_ROWS_NT=0: ordinary stores; 2, 3, 4: other cache
 * policies (rt_march.h), for measurements */
#ifndef RT_ROWS_NT
#define RT_ROWS_NT 1
#endif
/* the launch rows are read once per trace (-DRT_INPUT_NT=1: non-temporal
 * loads, measured in round 4) */
#if defined(RT_INPUT_NT) && RT_INPUT_NT
#define RT_INPUT_LOAD(p) __builtin_nontemporal_load(p)
#else
#define RT_INPUT_LOAD(p) (*(p))
#endif

/*
 * THE kernel: one lane owns one ray, state in VGPRs across the whole surface
 * loop, surface table through scalar loads, 7-10 coalesced 512-byte stores
 * per wavefront and element.
 */
/*
 * Launch rays whose components are the SAME bit pattern across the 64 rays of
 * a wavefront are not read 64 times.  The seed kernels keep, per 64-ray tile
 * of row 0, a note -- bit c: component c of y0 y1 y2 u0 u1 u2 is uniform --
 * and the tile's first ray, packed: rt_tiles { note[tiles], first[6][tiles] }.
 * The trace takes a uniform component from `first` (8 B per tile, sixteen
 * tiles to a cache line, read through the scalar path) instead of a 512-byte
 * segment of the row.  Collimated bundles (a field point at infinity: one
 * direction, rays starting on a plane) read 16 instead of 48 B per ray,
 * bundles from an object point 24; the values, hence the results, are the
 * same bits.  Note and index are wave-uniform (SGPRs): nothing diverges.
 */
/*
 * A direction's third component is redundant where it IS the completion of
 * the other two (the reference completes two-component directions itself:
 * u_z = sqrt(1 - (u_x^2 + u_y^2)), rayopt/geometric_trace.py:57-60; a caller's
 * own generator usually writes sqrt(1 - u_x^2 - u_y^2)).  The seed kernels
 * evaluate both forms with the device's own operations and note, per 64-ray
 * tile, whether every stored u_z equals one of them BIT FOR BIT (bits 6 and
 * 7 of the tile's note); the trace then rebuilds u_z of such a tile with the
 * same function instead of reading 8 B per ray among the saturated stores.
 * Same values by construction; -u_z, NaN rays, anything else: read.
 */
#define RT_NOTE_UZ_A 64u  /* u_z == sqrt((1 - u_x^2) - u_y^2) */
#define RT_NOTE_UZ_B 128u /* u_z == sqrt(1 - (u_x^2 + u_y^2)) */

__device__ __forceinline__ double rt_complete_uz(double ux, double uy, bool b)
{
    return b ? sqrt(1. - (ux * ux + uy * uy)) : sqrt((1. - ux * ux) - uy * uy);
}

__device__ __forceinline__ unsigned rt_same_bits(double v, double w,
                                                 unsigned bit)
{
    return __ballot(__double_as_longlong(v) != __double_as_longlong(w)) == 0ull
               ? bit : 0u;
}

struct rt_tiles {
    unsigned *note;   /* [tiles], already offset to the launch's first tile */
    double *first;    /* [6][stride], likewise */
    int64_t stride;   /* tiles of the whole batch */
};

__device__ __forceinline__ void rt_load_state_tiles(
    const rt_lay &a, int srow, int64_t col, const rt_tiles &tl, int tile,
    double (&y)[1][3], double (&u)[1][3])
{
    const unsigned m = tl.note[tile];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if ((m >> c) & 1)
            y[0][c] = tl.first[c * tl.stride + tile];
        else
            y[0][c] = RT_INPUT_LOAD(&a.Y[srow * a.ss + c * a.cs + col]);
        if ((m >> (3 + c)) & 1)
            u[0][c] = tl.first[(3 + c) * tl.stride + tile];
        else if (c == 2 && (m & (RT_NOTE_UZ_A | RT_NOTE_UZ_B)))
            u[0][2] = rt_complete_uz(u[0][0], u[0][1], !(m & RT_NOTE_UZ_A));
        else
            u[0][c] = RT_INPUT_LOAD(&a.U[srow * a.ss + c * a.cs + col]);
    }
}

template <bool ASPH>
__global__ void __launch_bounds__(RT_BLOCK)
rt_trace_kernel(const rt_surface *__restrict__ surf, int start, int stop,
                int clip, rt_lay a, int64_t ld, int64_t group_rays, int nsurf,
                int ngroups, rt_tiles tiles)
{
    const int64_t j = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (j >= ld)
        return;
    if (group_rays) {
        /* ray groups with their own surface table (one wavelength each):
         * group boundaries are multiples of 64 rays, so the group -- and
         * with it every table read -- stays wave-uniform (SGPRs) */
        const int64_t j0 = a.j0 + j - (int64_t)(threadIdx.x & 63);
        const int g = __builtin_amdgcn_readfirstlane((int)(j0 / group_rays));
        /* wavefronts of padding slots beyond the last ray (a batch in blocks
         * is padded to whole workgroups per block) take the last table: there
         * is none behind it */
        surf += (int64_t)(g < ngroups ? g : ngroups - 1) * nsurf;
    }
    const int64_t col = rt_col_wg(a, j, blockIdx.x);
    double y[1][3], u[1][3];
    if (tiles.note) {
        const int tile = __builtin_amdgcn_readfirstlane(
            (int)((j - (threadIdx.x & 63)) >> 6));
        rt_load_state_tiles(a, start - 1, col, tiles, tile, y, u);
    } else {
        rt_load_state<1>(a, start - 1, col, y, u);
    }
    rt_march<1, RT_ROWS_NT, ASPH>(surf, start, stop, clip, a, col, y, u);
}

/*
 * MEASUREMENT, not on the path (rt_newton_census): the same march from row
 * start - 1 with nothing stored, the Newton solves of the aspheric elements
 * counting -- per wavefront its trips through the iteration, per lane the
 * iterates its own ray needed.  out[0] += 64 x trips (lane slots the hardware
 * spent), out[1] += iterates (lane slots that did work), out[2] += trips,
 * out[3] += solves a wavefront entered with a live ray.
 */
__global__ void __launch_bounds__(RT_BLOCK)
rt_census_kernel(const rt_surface *__restrict__ surf, int start, int stop,
                 int clip, rt_lay a, int64_t ld, int64_t n, int64_t group_rays,
                 int nsurf, int ngroups, unsigned long long *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * RT_BLOCK + threadIdx.x;
    if (j >= ld)
        return;
    if (group_rays) {
        const int64_t j0 = a.j0 + j - (int64_t)(threadIdx.x & 63);
        const int g = __builtin_amdgcn_readfirstlane((int)(j0 / group_rays));
        surf += (int64_t)(g < ngroups ? g : ngroups - 1) * nsurf;
    }
    const int64_t col = rt_col_wg(a, j, blockIdx.x);
    double y[1][3], u[1][3], iv[1][3], t[1];
    rt_load_state<1>(a, start - 1, col, y, u);
    if (j >= n) /* padding slots (zeros in row 0) are no rays */
        u[0][0] = u[0][1] = u[0][2] = RT_NAN;
    unsigned census[3] = {0u, 0u, 0u};
    {
        const rt_surface *S0 = surf + (start - 1);
        rt_leave<1>(S0, S0->flags, y, u);
    }
    for (int s = start; s < stop; ++s) {
        const rt_surface *S = surf + s;
        const unsigned flags = S->flags;
        if (RT_WAVE_ANY(u[0][0] == u[0][0])) {
            rt_step_hit<1, true>(S, flags, y, u, iv, t, census);
            rt_step_bend<1>(S, flags, clip, y, iv, u);
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                y[0][c] = u[0][c] = RT_NAN;
        }
        rt_leave<1>(S, flags, y, u);
    }
    unsigned mine = j < n ? census[1] : 0u; /* (padding slots: no rays) */
    for (int off = 32; off > 0; off >>= 1)
        mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && census[0]) {
        atomicAdd(out + 0, 64ull * census[0]);
        atomicAdd(out + 1, (unsigned long long)mine);
        atomicAdd(out + 2, (unsigned long long)census[0]);
        atomicAdd(out + 3, (unsigned long long)census[2]);
    }
}

/*
 * The first trace after rt_generate_rays: the launch rays are built in
 * registers (field f = j / npupil through pupil point j % npupil), row 0 is
 * written from there and the march goes on -- the generated batch never
 * makes the round trip through HBM that a separate generation kernel plus
 * the 48 B/ray input read would cost (the read is the expensive kind:
 * profiles/r01_probes/ab_store_order.log (9)).  Later traces of the same
 * batch from element 1 (store0 = 0) build the rays again the same way --
 * the values row 0 holds, bit for bit -- instead of reading them.
 */
template <bool ASPH>
__global__ void __launch_bounds__(RT_BLOCK)
rt_trace_gen_kernel(const rt_surface *__restrict__ surf, int stop, int clip,
                    rt_lay a, int64_t ld, int64_t group_rays, int nsurf,
                    int ngroups, const rt_field *__restrict__ fields,
                    const double *__restrict__ pupil, int64_t npupil,
                    int64_t n, int64_t j0, rt_surface S0, int store_i0,
                    int store0, rt_gen_order order)
{
    /* `a` and `ld` describe the window of columns this launch covers (the
     * whole batch: j0 = 0); j0 + column = index of the ray in the batch.
     * Large pupils are taken in turns (rt_gen_wg, rt_lay.h): wg = which 256
     * rays this workgroup has, wave-uniform */
    const uint32_t wg = rt_gen_wg(order, blockIdx.x);
    const int64_t w = (int64_t)wg * RT_BLOCK + threadIdx.x;
    if (w >= ld)
        return;
    const int64_t j = j0 + w;
    if (group_rays) {
        const int64_t jw = j - (int64_t)(threadIdx.x & 63);
        const int g = __builtin_amdgcn_readfirstlane((int)(jw / group_rays));
        surf += (int64_t)(g < ngroups ? g : ngroups - 1) * nsurf;
    }
    double y[1][3] = {{0., 0., 0.}}, u[1][3] = {{0., 0., 0.}};
    if (j < n) {
        const int64_t p = j % npupil;
        /* (ASPH = false: the first element is no asphere either) */
        rt_generate_ray<ASPH>(fields + j / npupil, pupil[2 * p],
                              pupil[2 * p + 1], &S0, y, u);
    }
    const int64_t col = rt_col_wg(a, w, wg);
    if (store0) { /* first trace of the batch; later ones leave row 0 alone */
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.Y[c * a.cs + col] = y[0][c];
            a.U[c * a.cs + col] = u[0][c];
            if (store_i0)
                a.I[c * a.cs + col] = u[0][c];
        }
        a.T[col] = 0.;
    }
    rt_march<1, RT_ROWS_NT, ASPH>(surf, 1, stop, clip, a, col, y, u);
}

/*
 * Clipped-ray compaction (BASELINE north_star: "wavefront ballots for ...
 * clipped-ray compaction").  A ray whose direction has become NaN -- clipped
 * by an aperture (rayopt/elements.py:206-209), missed surface, TIR, Newton
 * failure -- is NaN in every array of every later element, so there is
 * nothing left to compute for it; with rows that are all stored the kernel is
 * bound by those stores and a dead ray costs exactly what a live one does
 * (measured: profiles/r02_probes, part C), but when rows are NOT stored
 * (rt_set_keep_rows: merit functions keep the image row) the kernel is bound
 * by FP64 issue and dead lanes are wasted issue slots.  This variant retires
 * dead rays -- their remaining kept rows are filled with NaN at once -- and,
 * whenever that frees a whole wavefront of the 256-ray workgroup, packs the
 * surviving rays into the low lanes: 64-bit ballots + popcounts give every
 * survivor its slot, the state (y, u and the ray's column) moves through LDS,
 * and the emptied wavefronts only keep the barriers company.  A ray keeps
 * its column, so results land where the plain kernel puts them, bit for bit.
 */
#define RT_CB 256

/* NaN into the rows of element s for one column (flags say which exist) */
__device__ __forceinline__ void rt_store_nan_row(unsigned f, int s,
                                                 const rt_lay &a, int64_t col)
{
    const int64_t row = s * a.ss + col;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.Y[row + c * a.cs] = RT_NAN;
        if (!(f & RT_F_SKIP_U))
            a.U[row + c * a.cs] = RT_NAN;
        if (f & RT_F_STORE_I)
            a.I[row + c * a.cs] = RT_NAN;
    }
    a.T[s * a.ssT + col] = RT_NAN;
}

__global__ void __launch_bounds__(RT_CB)
rt_trace_compact_kernel(const rt_surface *__restrict__ surf, int start,
                        int stop, int clip, rt_lay a, int64_t ld,
                        int64_t group_rays, int nsurf, int ngroups,
                        int every)
{
    __shared__ int cnt[2][RT_CB / 64]; /* by parity of the ROUND (the k-th
                                          time the question is asked): a
                                          wavefront may still read round k
                                          while another already writes round
                                          k + 1 */
    __shared__ double sm[6][RT_CB];
    __shared__ int smi[RT_CB];
    __shared__ unsigned short gone[RT_CB]; /* column -> first element whose
                                              rows are NaN (0: alive) */
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t tile0 = (int64_t)blockIdx.x * RT_CB;
    if (group_rays) { /* a tile never straddles two groups (host checks) */
        const int64_t g = tile0 / group_rays;
        surf += (g < ngroups ? g : ngroups - 1) * nsurf;
    }
    /* SoA (the only layout this kernel is launched for): the tile's columns
     * are consecutive -- blocks are whole tiles (rt_reserve) */
    const int64_t col0 = rt_col_wg(a, tile0, blockIdx.x);
    const bool exists = tile0 + tid < ld;
    bool has = exists;
    int idx = tid; /* the ray's column inside the tile */
    gone[tid] = 0;
    double y[1][3] = {{0., 0., 0.}}, u[1][3] = {{0., 0., 0.}};
    if (has)
        rt_load_state<1>(a, start - 1, col0 + idx, y, u);
    {
        const rt_surface *S0 = surf + (start - 1);
        rt_leave<1>(S0, S0->flags, y, u);
    }
    int nwaves = RT_CB / 64; /* wavefronts that may still hold rays */
    for (int s = start; s < stop; ++s) {
        /* retire the rays that died at the previous element: their later
         * kept rows are NaN, written below when those rows come up */
        if (has && !(u[0][0] == u[0][0])) {
            gone[idx] = (unsigned short)s; /* a wavefront that is still at
                                              the rows of element s-1 reads
                                              "not yet" */
            has = false;
        }
        /* survivors per wavefront -> can a whole wavefront be freed?  The
         * question costs a workgroup barrier, so it is asked only at every
         * `every`-th element (uniform across the workgroup) */
        const bool ask = (s - start) % every == every - 1 && nwaves > 1;
        if (ask) {
            const int par = ((s - start) / every) & 1;
            const unsigned long long mine = __ballot(has);
            if (wave < nwaves && lane == 0)
                cnt[par][wave] = __popcll(mine);
            __syncthreads();
            int total = 0, before = 0, used = 0;
            for (int w = 0; w < nwaves; ++w) {
                const int c = cnt[par][w];
                before += w < wave ? c : 0;
                total += c;
                used += c > 0;
            }
            const int need = (total + 63) >> 6;
            if (need < used) { /* workgroup-uniform */
                if (has) {
                    const int dst =
                        before + __popcll(mine & ((1ull << lane) - 1ull));
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        sm[c][dst] = y[0][c];
                        sm[3 + c][dst] = u[0][c];
                    }
                    smi[dst] = idx;
                }
                __syncthreads();
                has = tid < total;
                if (has) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        y[0][c] = sm[c][tid];
                        u[0][c] = sm[3 + c][tid];
                    }
                    idx = smi[tid];
                }
                nwaves = need;
                __syncthreads(); /* sm is rewritten by the next compaction */
            }
        }
        const rt_surface *S = surf + s;
        const unsigned flags = S->flags;
        if (RT_WAVE_ANY(has)) {
            double iv[1][3], t[1];
            rt_step<1>(S, flags, clip, y, u, iv, t);
            if (has)
                rt_store_rows<1, RT_ROWS_NT>(flags, s, a, col0 + idx, y, u, iv, t);
            rt_leave<1>(S, flags, y, u);
        }
        if (!(flags & RT_F_NOSTORE)) {
            /* a kept row: the columns of retired rays get their NaN from the
             * thread that owns the column, next to the survivors' stores */
            __syncthreads();
            const int from = gone[tid];
            if (exists && from && from <= s)
                rt_store_nan_row(flags, s, a, col0 + tid);
        }
    }
}

/* bit c (Y) / 3+c (U) of the tile's note: all 64 rays of the wavefront hold
 * the same bit pattern in that component */
__device__ __forceinline__ unsigned rt_uniform_bit(double v, int bit)
{
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_readfirstlane((int)b);
    const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
    const long long first = ((long long)hi << 32) | (unsigned)lo;
    return __ballot(b != first) == 0ull ? 1u << bit : 0u;
}

/* rays_given: AoS (n,3) staging -> SoA row 0 of Y,U,I and T[0] = 0 */
__global__ void rt_seed_aos_kernel(const double *__restrict__ y_aos,
                                   const double *__restrict__ u_aos,
                                   int64_t n, rt_lay a, int64_t ld,
                                   int store_i, int64_t period,
                                   rt_tiles tiles, int64_t j0, int64_t j1)
{
    /* rays j0 .. j1 - 1 of the batch (a window of whole workgroups: an
     * upload is seeded window by window while the next one crosses PCIe) */
    const int64_t j = j0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld || j >= j1)
        return;
    const bool in = j < n;
    const int64_t k = j % period; /* the same rays for every group */
    const int64_t col = rt_col(a, j);
    unsigned note = 0;
    double first[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double p = in ? y_aos[k * 3 + c] : 0.;
        const double q = in ? u_aos[k * 3 + c] : 0.;
        first[c] = p;
        first[3 + c] = q;
        a.Y[c * a.cs + col] = p;
        a.U[c * a.cs + col] = q;
        if (store_i)
            a.I[c * a.cs + col] = q;
        note |= rt_uniform_bit(p, c) | rt_uniform_bit(q, 3 + c);
    }
    note |= rt_same_bits(first[5], rt_complete_uz(first[3], first[4], false),
                         RT_NOTE_UZ_A) |
            rt_same_bits(first[5], rt_complete_uz(first[3], first[4], true),
                         RT_NOTE_UZ_B);
    a.T[col] = 0.;
    if (tiles.note && (threadIdx.x & 63) == 0) {
        /* (ld is a multiple of 64: a wavefront is one tile; a tile with
         * padding columns beyond n is read the ordinary way) */
        const int64_t t = j >> 6;
        tiles.note[t] = j + 64 <= n ? note : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tiles.first[c * tiles.stride + t] = first[c];
            tiles.first[(3 + c) * tiles.stride + t] = first[3 + c];
        }
    }
}

/* rays_given for SoA (3,n) device/staged input */
__global__ void rt_seed_soa_kernel(const double *__restrict__ y_soa,
                                   const double *__restrict__ u_soa,
                                   int64_t n, rt_lay a, int64_t ld,
                                   int store_i, int64_t period,
                                   rt_tiles tiles, int64_t j0, int64_t j1)
{
    /* rays j0 .. j1 - 1 of the batch (a window of whole workgroups: an
     * upload is seeded window by window while the next one crosses PCIe) */
    const int64_t j = j0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld || j >= j1)
        return;
    const bool in = j < n;
    const int64_t k = j % period; /* the same rays for every group */
    const int64_t col = rt_col(a, j);
    unsigned note = 0;
    double first[6];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double p = in ? y_soa[c * period + k] : 0.;
        const double q = in ? u_soa[c * period + k] : 0.;
        first[c] = p;
        first[3 + c] = q;
        a.Y[c * a.cs + col] = p;
        a.U[c * a.cs + col] = q;
        if (store_i)
            a.I[c * a.cs + col] = q;
        note |= rt_uniform_bit(p, c) | rt_uniform_bit(q, 3 + c);
    }
    note |= rt_same_bits(first[5], rt_complete_uz(first[3], first[4], false),
                         RT_NOTE_UZ_A) |
            rt_same_bits(first[5], rt_complete_uz(first[3], first[4], true),
                         RT_NOTE_UZ_B);
    a.T[col] = 0.;
    if (tiles.note && (threadIdx.x & 63) == 0) {
        /* (ld is a multiple of 64: a wavefront is one tile; a tile with
         * padding columns beyond n is read the ordinary way) */
        const int64_t t = j >> 6;
        tiles.note[t] = j + 64 <= n ? note : 0u;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            tiles.first[c * tiles.stride + t] = first[c];
            tiles.first[(3 + c) * tiles.stride + t] = first[3 + c];
        }
    }
}

/* rays of field f x pupil point p, see rt_generate_rays in rt_mi355.h */
__global__ void rt_generate_kernel(const rt_field *__restrict__ fields,
                                   const double *__restrict__ pupil,
                                   int64_t npupil, int64_t n, rt_surface S0,
                                   rt_lay a, int64_t ld, int store_i)
{
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= ld)
        return;
    double y[1][3] = {{0., 0., 0.}}, u[1][3] = {{0., 0., 0.}};
    if (r < n) {
        const int64_t p = r % npupil;
        rt_generate_ray(fields + r / npupil, pupil[2 * p], pupil[2 * p + 1],
                        &S0, y, u);
    }
    const int64_t col = rt_col(a, r);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        a.Y[c * a.cs + col] = y[0][c];
        a.U[c * a.cs + col] = u[0][c];
        if (store_i)
            a.I[c * a.cs + col] = u[0][c];
    }
    a.T[col] = 0.;
}

/*
 * After a surface table has been copied to the device: the refined
 * reciprocal of every curved element's c, formed by the device's own
 * division sequence (rt_rcp_refined, rt_math.h).  One thread per element.
 */
__global__ void rt_table_finish_kernel(rt_surface *__restrict__ tab, int ntab)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < ntab)
        tab[j].rc = (tab[j].flags & RT_F_CURVED) ? rt_rcp_refined(tab[j].c) : 0.;
}

/*
 * rt_selftest_arith: the quotient and the square root without the range
 * scaffolding against the compiler's sequences, bit for bit, on operands
 * drawn from a counter-based generator: magnitudes 2^e with e uniform in
 * [-span, span], random mantissas and signs, plus the edge values (0, inf,
 * NaN, denormals, the guard limits and their neighbours) every 64th draw.
 * The guarded forms are exactly the ones rt_math.h uses: outside the range
 * they fall back, so they must agree EVERYWHERE; `raw` counts how often the
 * unguarded core differs (expected > 0 only when span > 100).
 */
__device__ __forceinline__ unsigned long long rt_mix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ double rt_draw(unsigned long long key, int span)
{
    const unsigned long long h = rt_mix64(key);
    if ((h & 63ull) == 0ull) {
        const double edge[16] = {0., -0., __builtin_inf(), -__builtin_inf(),
                                 __builtin_nan(""), 0x1p-1074, 0x1p-1022,
                                 RT_RANGE_BIG, RT_RANGE_TINY, 0x1p-767,
                                 0x1.fffffffffffffp99, 0x1.fffffffffffffp-101,
                                 1., 0x1.fffffffffffffp-1, 0x1p1023, 0x1p-969};
        return edge[(h >> 6) & 15ull];
    }
    const int e = (int)((h >> 12) % (unsigned long long)(2 * span + 1)) - span;
    const unsigned long long bits =
        ((h >> 63) << 63) | ((unsigned long long)(1023 + e) << 52) |
        (rt_mix64(h) >> 12);
    return __longlong_as_double((long long)bits);
}

__device__ __forceinline__ bool rt_same_bits(double a, double b)
{
    return __double_as_longlong(a) == __double_as_longlong(b) ||
           (a != a && b != b);
}

__global__ void rt_selftest_kernel(unsigned long long seed, long long n,
                                   int span, unsigned long long *counts)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long bad_q = 0, bad_c = 0, bad_s = 0, raw = 0;
    if (i < n) {
        const unsigned long long k = seed * 0x100000001B3ull + 3ull * i;
        const double num = rt_draw(k, span);
        double den = rt_draw(k + 1, span);
        const double x = rt_draw(k + 2, span);
        /* (1) two quotients sharing r2-like denominators >= 1 (refraction) */
        const double r2 = __builtin_fabs(den) + 1.;
        {
            const double want = num / r2;
            double got;
            if (RT_WAVE_ANY(r2 >= RT_RANGE_BIG || RT_ODD_MAG(num)))
                got = num / r2;
            else
                got = rt_quot(num, r2, rt_rcp_refined(r2));
            bad_q += !rt_same_bits(want, got);
            raw += !rt_same_bits(want, rt_quot(num, r2, rt_rcp_refined(r2)));
        }
        /* (2) a quotient by a table constant (the sphere's c) */
        {
            const double want = num / den;
            double got;
            if (RT_ODD_MAG(den) || RT_WAVE_ANY(RT_ODD_MAG(num)))
                got = num / den;
            else
                got = rt_quot(num, den, rt_rcp_refined(den));
            bad_c += !rt_same_bits(want, got);
        }
        /* (3) the square root */
        {
            const double want = sqrt(x);
            double got;
            if (RT_WAVE_ANY(x > 0. && x < RT_SQRT_SCALED_BELOW))
                got = sqrt(x);
            else
                got = rt_sqrt_unscaled(x);
            bad_s += !rt_same_bits(want, got);
        }
    }
    if (bad_q)
        atomicAdd(&counts[0], bad_q);
    if (bad_c)
        atomicAdd(&counts[1], bad_c);
    if (bad_s)
        atomicAdd(&counts[2], bad_s);
    if (raw)
        atomicAdd(&counts[3], raw);
}

#endif /* RT_TRACE_KERNELS_H */
