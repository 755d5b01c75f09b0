/*
 * rt_kernels.h -- every piece of device code of the engine: the fused trace
 * kernel and its helpers, ray seeding / generation, the bandwidth probes and
 * the reductions behind rms / refocus / opd / resize.  Included once by
 * rt_engine.hip (host side: contexts, C ABI, RCCL); the per-ray arithmetic
 * itself lives in rt_math.h.
 */
#ifndef RT_KERNELS_H
#define RT_KERNELS_H

#include <hip/hip_runtime.h>
#include "rt_math.h"
#include "rt_aim.h"

/* ------------------------------------------------------------------ */
/* kernels                                                            */
/* ------------------------------------------------------------------ */

template <int R> struct rt_vec;
template <> struct rt_vec<1> { typedef double type; };
template <> struct rt_vec<2> {
    typedef double type __attribute__((ext_vector_type(2)));
};
template <> struct rt_vec<4> {
    typedef double type __attribute__((ext_vector_type(4)));
};

template <int R>
__device__ __forceinline__ void rt_load(const double *__restrict__ p,
                                        double (&v)[R])
{
    typedef typename rt_vec<R>::type V;
    const V x = *reinterpret_cast<const V *>(p);
    if constexpr (R == 1) {
        v[0] = x;
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            v[r] = x[r];
    }
}

template <int R, bool NT>
__device__ __forceinline__ void rt_store(double *__restrict__ p,
                                         const double (&v)[R])
{
    typedef typename rt_vec<R>::type V;
    V x;
    if constexpr (R == 1) {
        x = v[0];
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            x[r] = v[r];
    }
    if constexpr (NT)
        __builtin_nontemporal_store(x, reinterpret_cast<V *>(p));
    else
        *reinterpret_cast<V *>(p) = x;
}

/*
 * blockIdx -> chunk of rays.  Workgroup b is dispatched to XCD b % 8
 * (observed, used for speed only).  With XCD = true the chunks are dealt so
 * that each XCD streams one contiguous eighth of every result row instead of
 * every eighth 4 KiB chunk.
 */
template <bool XCD>
__device__ __forceinline__ int64_t rt_chunk(int64_t nblocks)
{
    const int64_t b = blockIdx.x;
    if constexpr (!XCD)
        return b;
    const int64_t per = (nblocks + 7) / 8;
    const int64_t c = (b & 7) * per + (b >> 3);
    return c; /* may be >= nblocks for the ragged tail: caller checks */
}

/*
 * Where the result arrays live.  Default ("SoA", include/rt_mi355.h):
 *     Y,U,I [L][3][ld], T [L][ld]      cs = ld, ss = 3 ld, ssT = ld
 * and a ray's column is its index j.  Measurement-only alternative
 * (rt_set_option "tile_rays" = TR): the batch is cut into tiles of TR rays
 * and a tile holds ALL its rows back to back, [tile][L][10][TR] with the ten
 * components y0 y1 y2 u0 u1 u2 i0 i1 i2 t, so that a workgroup's whole
 * output is one contiguous region:
 *     cs = TR, ss = ssT = 10 TR, tile stride ts = L 10 TR.
 * Element (array, s, c) of ray j is at
 *     base[array] + s*ss + c*cs + (j >> tshift)*ts + (j & (TR - 1)).
 * For SoA ts = TR = 1 << tshift, which makes the last two terms j again.
 */
struct rt_lay {
    double *Y, *U, *I, *T;
    int64_t cs, ss, ssT, ts;
    int tshift;
};

__device__ __forceinline__ int64_t rt_col(const rt_lay &a, int64_t j)
{
    return (j >> a.tshift) * a.ts + (j & (((int64_t)1 << a.tshift) - 1));
}

/* read rows start-1 of Y,U for the R rays at column `col` */
template <int R>
__device__ __forceinline__ void rt_load_state(const rt_lay &a, int srow,
                                              int64_t col, double (&y)[R][3],
                                              double (&u)[R][3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p[R], q[R];
        rt_load<R>(a.Y + srow * a.ss + c * a.cs + col, p);
        rt_load<R>(a.U + srow * a.ss + c * a.cs + col, q);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            y[r][c] = p[r];
            u[r][c] = q[r];
        }
    }
}

/*
 * The same for one ray per lane where some components of the input rows are
 * known to hold ONE bit pattern across the wavefront's 64 rays (bit c of
 * `m`: Y component c, bit 3+c: U component c): those are read from the
 * wavefront's first column -- one request instead of eight cache lines.
 * Bundles from a field point at infinity share their direction, bundles from
 * an object point share their origin (rayopt/conjugates.py:137-166,236-255),
 * so half or more of the 48 B/ray input is of this kind.
 */
__device__ __forceinline__ void rt_load_state_uniform(
    const rt_lay &a, int srow, int64_t col, int64_t col0, unsigned m,
    double (&y)[1][3], double (&u)[1][3])
{
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double *py = a.Y + srow * a.ss + c * a.cs;
        const double *pu = a.U + srow * a.ss + c * a.cs;
        y[0][c] = py[(m >> c) & 1 ? col0 : col];
        u[0][c] = pu[(m >> (3 + c)) & 1 ? col0 : col];
    }
}

/* the rows of one element for the R rays at column `col` */
template <int R, bool NT>
__device__ __forceinline__ void rt_store_rows(
    unsigned flags, int s, const rt_lay &a, int64_t col,
    const double (&y)[R][3], const double (&u)[R][3],
    const double (&iv)[R][3], const double (&t)[R])
{
    if (flags & RT_F_NOSTORE)
        return;
    const int64_t row = s * a.ss + col;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p[R], q[R], d[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            p[r] = y[r][c];
            q[r] = u[r][c];
            d[r] = iv[r][c];
        }
        rt_store<R, NT>(a.Y + row + c * a.cs, p);
        if (!(flags & RT_F_SKIP_U))
            rt_store<R, NT>(a.U + row + c * a.cs, q);
        if (flags & RT_F_STORE_I)
            rt_store<R, NT>(a.I + row + c * a.cs, d);
    }
    rt_store<R, NT>(a.T + s * a.ssT + col, t);
}

/* all elements start..stop-1 for the R rays of this lane; state in VGPRs */
template <int R, bool NT>
__device__ __forceinline__ void rt_march(const rt_surface *__restrict__ surf,
                                         int start, int stop, int clip,
                                         const rt_lay &a, int64_t col,
                                         double (&y)[R][3],
                                         double (&u)[R][3])
{
    double iv[R][3], t[R];
    {
        const rt_surface *S0 = surf + (start - 1);
        rt_leave<R>(S0, S0->flags, y, u);
    }
    for (int s = start; s < stop; ++s) {
        const rt_surface *S = surf + s;
        const unsigned flags = S->flags;
        /* a ray whose direction is NaN (clipped, missed, TIR, Newton failure:
         * elements.py:206-209,:496,:367,:347) yields NaN in every array of
         * every later element; a wavefront with no other ray left stores
         * that without evaluating it */
        bool alive = false;
#pragma unroll
        for (int r = 0; r < R; ++r)
            alive = alive || u[r][0] == u[r][0];
        if (RT_WAVE_ANY(alive)) {
            rt_step<R>(S, flags, clip, y, u, iv, t);
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[r] = RT_NAN;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    y[r][c] = u[r][c] = iv[r][c] = RT_NAN;
            }
        }

        /* all rows of the element leave in one burst: measured 3 % faster
         * than sending y,t,i ahead of the refraction, and aligning the waves
         * of a workgroup with a barrier first does not help
         * (profiles/r01_probes/ab_store_order.log) */
        rt_store_rows<R, NT>(flags, s, a, col, y, u, iv, t);

        rt_leave<R>(S, flags, y, u);
    }
}

template <int R, bool NT, bool XCD>
__global__ void rt_trace_kernel(const rt_surface *__restrict__ surf, int start,
                                int stop, int clip, rt_lay a, int64_t ld,
                                int64_t nblocks, int64_t group_rays,
                                int nsurf, const unsigned *__restrict__ uni,
                                unsigned ufix, unsigned gate_mask,
                                unsigned gate_window)
{
    const int64_t chunk = rt_chunk<XCD>(nblocks);
    const int64_t j = (chunk * blockDim.x + threadIdx.x) * R;
    if (j >= ld)
        return;
    if (group_rays) {
        /* ray groups with their own surface table (one wavelength each):
         * group boundaries are multiples of 64 R rays, so the group -- and
         * with it every table read -- stays wave-uniform (SGPRs) */
        const int64_t j0 = j - (int64_t)(threadIdx.x & 63) * R;
        const int g = __builtin_amdgcn_readfirstlane((int)(j0 / group_rays));
        surf += (int64_t)g * nsurf;
    }
    const int64_t col = rt_col(a, j);
    if (gate_mask) {
        /* measurement: input reads only inside chip-wide time windows (the
         * 100 MHz reference counter is the same on every CU) */
        while (((unsigned)__builtin_amdgcn_s_memrealtime() & gate_mask) >=
               gate_window)
            __builtin_amdgcn_s_sleep(2);
    }
    double y[R][3], u[R][3];
    if constexpr (R == 1) {
        if (uni || ufix) {
            /* per 64-ray tile: which input components are wave-uniform */
            const int tile = __builtin_amdgcn_readfirstlane(
                (int)((j - (int64_t)(threadIdx.x & 63)) >> 6));
            const unsigned m = uni ? uni[tile] : ufix;
            rt_load_state_uniform(a, start - 1, col,
                                  rt_col(a, (int64_t)tile << 6), m, y, u);
        } else {
            rt_load_state<R>(a, start - 1, col, y, u);
        }
    } else {
        rt_load_state<R>(a, start - 1, col, y, u);
    }
    rt_march<R, NT>(surf, start, stop, clip, a, col, y, u);
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
__global__ void rt_trace_gen_kernel(const rt_surface *__restrict__ surf,
                                    int stop, int clip, rt_lay a, int64_t ld,
                                    int64_t group_rays, int nsurf,
                                    const rt_field *__restrict__ fields,
                                    const double *__restrict__ pupil,
                                    int64_t npupil, int64_t n, rt_surface S0,
                                    int store_i0, int store0)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld)
        return;
    if (group_rays) {
        const int64_t j0 = j - (int64_t)(threadIdx.x & 63);
        const int g = __builtin_amdgcn_readfirstlane((int)(j0 / group_rays));
        surf += (int64_t)g * nsurf;
    }
    double y[1][3] = {{0., 0., 0.}}, u[1][3] = {{0., 0., 0.}};
    if (j < n) {
        const int64_t p = j % npupil;
        rt_generate_ray(fields + j / npupil, pupil[2 * p], pupil[2 * p + 1],
                        &S0, y, u);
    }
    const int64_t col = rt_col(a, j);
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
    rt_march<1, false>(surf, 1, stop, clip, a, col, y, u);
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
                        int64_t group_rays, int nsurf, int every)
{
    __shared__ int cnt[2][RT_CB / 64]; /* by element parity: a wavefront may
                                          still read round k while another
                                          already writes round k + 1 */
    __shared__ double sm[6][RT_CB];
    __shared__ int smi[RT_CB];
    __shared__ unsigned short gone[RT_CB]; /* column -> first element whose
                                              rows are NaN (0: alive) */
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int64_t tile0 = (int64_t)blockIdx.x * RT_CB;
    if (group_rays) /* a tile never straddles two groups (host checks) */
        surf += (tile0 / group_rays) * nsurf;
    /* SoA (the only layout this kernel is launched for): the tile's columns
     * are consecutive */
    const int64_t col0 = tile0;
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
            const unsigned long long mine = __ballot(has);
            if (wave < nwaves && lane == 0)
                cnt[s & 1][wave] = __popcll(mine);
            __syncthreads();
            int total = 0, before = 0, used = 0;
            for (int w = 0; w < nwaves; ++w) {
                const int c = cnt[s & 1][w];
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
                rt_store_rows<1, false>(flags, s, a, col0 + idx, y, u, iv, t);
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

/* rays_given: AoS (n,3) staging -> SoA row 0 of Y,U,I and T[0] = 0 */
__global__ void rt_seed_aos_kernel(const double *__restrict__ y_aos,
                                   const double *__restrict__ u_aos,
                                   int64_t n, rt_lay a, int64_t ld,
                                   int store_i, int64_t period)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld)
        return;
    const bool in = j < n;
    const int64_t k = j % period; /* the same rays for every group */
    const int64_t col = rt_col(a, j);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double p = in ? y_aos[k * 3 + c] : 0.;
        const double q = in ? u_aos[k * 3 + c] : 0.;
        a.Y[c * a.cs + col] = p;
        a.U[c * a.cs + col] = q;
        if (store_i)
            a.I[c * a.cs + col] = q;
    }
    a.T[col] = 0.;
}

/* rays_given for SoA (3,n) device/staged input */
__global__ void rt_seed_soa_kernel(const double *__restrict__ y_soa,
                                   const double *__restrict__ u_soa,
                                   int64_t n, rt_lay a, int64_t ld,
                                   int store_i, int64_t period)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= ld)
        return;
    const bool in = j < n;
    const int64_t k = j % period; /* the same rays for every group */
    const int64_t col = rt_col(a, j);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const double p = in ? y_soa[c * period + k] : 0.;
        const double q = in ? u_soa[c * period + k] : 0.;
        a.Y[c * a.cs + col] = p;
        a.U[c * a.cs + col] = q;
        if (store_i)
            a.I[c * a.cs + col] = q;
    }
    a.T[col] = 0.;
}


/*
 * Bandwidth probes (measurement only): the store pattern of the trace kernel
 * without its arithmetic, a linear fill and a 16-byte copy.  They calibrate
 * the memory-system ceiling the trace kernel is judged against.
 */
/* store flavours of the pattern probe: 0 plain, 1 non-temporal, 2 sc1
 * (write-through to memory, line dropped from the XCD's L2), 3 sc0 sc1 */
template <int FL, typename V>
__device__ __forceinline__ void rt_probe_store(V *p, V v)
{
    if constexpr (FL == 0) {
        *p = v;
    } else if constexpr (FL == 1) {
        __builtin_nontemporal_store(v, p);
    } else if constexpr (sizeof(V) == 8) {
        if constexpr (FL == 2)
            asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p),
                         "v"(v)
                         : "memory");
        else
            asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p),
                         "v"(v)
                         : "memory");
    } else {
        if constexpr (FL == 2)
            asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p),
                         "v"(v)
                         : "memory");
        else
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p),
                         "v"(v)
                         : "memory");
    }
}

template <int IN, int RP, int FL>
__global__ void rt_probe_pattern_kernel(int start, int stop,
                                        const double *__restrict__ in,
                                        rt_lay a, int64_t ld, int stored_i)
{
    typedef typename rt_vec<RP>::type V;
    const int64_t j = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * RP;
    if (j >= ld)
        return;
    const int64_t col = rt_col(a, j);
    V y[3], u[3];
    for (int c = 0; c < 3; ++c) {
        if constexpr (IN == 0) { /* the 48 B/ray input rows, from HBM */
            y[c] = *reinterpret_cast<const V *>(
                a.Y + (start - 1) * a.ss + c * a.cs + col);
            u[c] = *reinterpret_cast<const V *>(
                a.U + (start - 1) * a.ss + c * a.cs + col);
        } else if constexpr (IN == 3) { /* the same, non-temporal loads */
            y[c] = __builtin_nontemporal_load(reinterpret_cast<const V *>(
                a.Y + (start - 1) * a.ss + c * a.cs + col));
            u[c] = __builtin_nontemporal_load(reinterpret_cast<const V *>(
                a.U + (start - 1) * a.ss + c * a.cs + col));
        } else if constexpr (IN == 4) { /* from a separate buffer `in`,
                                           [6][ld] (uncached allocation) */
            y[c] = *reinterpret_cast<const V *>(in + (int64_t)c * ld + j);
            u[c] = *reinterpret_cast<const V *>(in + (int64_t)(3 + c) * ld + j);
        } else if constexpr (IN == 1) { /* from a 3 MB window that stays in
                                           L2 */
            const int64_t k = j & 0xffff;
            y[c] = *reinterpret_cast<const V *>(in + (int64_t)c * 65536 + k);
            u[c] = *reinterpret_cast<const V *>(in + (int64_t)(3 + c) * 65536 +
                                                k);
        } else { /* no read at all */
            y[c] = (V)((double)threadIdx.x);
            u[c] = (V)((double)blockIdx.x);
        }
    }
    for (int s = start; s < stop; ++s) {
        const int64_t row = s * a.ss + col;
        for (int c = 0; c < 3; ++c) {
            y[c] += u[c];
            rt_probe_store<FL>(reinterpret_cast<V *>(a.Y + row + c * a.cs),
                               y[c]);
            rt_probe_store<FL>(reinterpret_cast<V *>(a.U + row + c * a.cs),
                               u[c]);
            if (stored_i)
                rt_probe_store<FL>(
                    reinterpret_cast<V *>(a.I + row + c * a.cs), u[c]);
        }
        rt_probe_store<FL>(reinterpret_cast<V *>(a.T + s * a.ssT + col),
                           y[2]);
    }
}

/* the 56 B pattern with K rays per lane marched ONE AFTER THE OTHER, all
 * K inputs loaded up front: K times fewer, K times larger read bursts per
 * workgroup (does clustering the reads in time make them cheaper among the
 * saturated writes?) */
template <int K>
__global__ void rt_probe_seq_kernel(int start, int stop, rt_lay a, int64_t ld)
{
    const int64_t base = (int64_t)blockIdx.x * blockDim.x * K + threadIdx.x;
    double y[K][3], u[K][3];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t j = base + (int64_t)k * blockDim.x;
        const int64_t col = rt_col(a, j < ld ? j : 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            y[k][c] = a.Y[(start - 1) * a.ss + c * a.cs + col];
            u[k][c] = a.U[(start - 1) * a.ss + c * a.cs + col];
        }
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int64_t j = base + (int64_t)k * blockDim.x;
        if (j >= ld)
            continue;
        const int64_t col = rt_col(a, j);
        for (int s = start; s < stop; ++s) {
            const int64_t row = s * a.ss + col;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                y[k][c] += u[k][c];
                a.Y[row + c * a.cs] = y[k][c];
                a.U[row + c * a.cs] = u[k][c];
            }
            a.T[s * a.ssT + col] = y[k][2];
        }
    }
}

__global__ void rt_probe_fill_kernel(double *__restrict__ dst, int64_t n2)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    v2 v = {1., 2.};
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
         i += stride)
        reinterpret_cast<v2 *>(dst)[i] = v;
}

template <bool NT>
__global__ void rt_probe_fill_once_kernel(double *__restrict__ dst, int64_t n2)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n2)
        return;
    v2 v = {1., 2.};
    if constexpr (NT)
        __builtin_nontemporal_store(v, reinterpret_cast<v2 *>(dst) + i);
    else
        reinterpret_cast<v2 *>(dst)[i] = v;
}

__global__ void rt_probe_copy_kernel(const double *__restrict__ src,
                                     double *__restrict__ dst, int64_t n2)
{
    typedef double v2 __attribute__((ext_vector_type(2)));
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2;
         i += stride)
        reinterpret_cast<v2 *>(dst)[i] = reinterpret_cast<const v2 *>(src)[i];
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
 * System.pupil for F fields: FOUR lanes per field.  Every lane of a field
 * repeats the chief-ray solve (same arithmetic, same result), then lane m
 * runs ONE of the four marginal root finds -- m = 0..3 in the order the
 * sequential rt_aim_field takes them (+mer, -mer, +sag, -sag) -- so the
 * longest dependent chain is chief + one marginal instead of chief + four.
 * The lanes then agree on what the sequential code would have returned: the
 * status of the first solve that failed, NaN for it and for every later
 * entry.  Bit-identical to rt_aim_field (tests/hostemu runs that one).
 *
 * UNIFORM: one field per workgroup (4 lanes of one wavefront): the field's
 * surface table is wave-uniform and is read with scalar loads, like the trace
 * kernel's -- the one-ray traces are chains of dependent table reads, and
 * per-lane vector loads of the table were what the solve waited for.
 * !UNIFORM: 16 fields per wavefront, for batches so large that one wavefront
 * per field would not be resident at once; the table is read per lane.
 */
template <bool UNIFORM>
__global__ void __launch_bounds__(64) rt_aim_kernel(const rt_surface *__restrict__ tab, int nsurf,
                              const rt_aim_seed *__restrict__ seeds, int nf,
                              rt_aim_args args, double *__restrict__ z,
                              double *__restrict__ a,
                              int32_t *__restrict__ status)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int f = t >> 2, m = t & 3;
    if (f >= nf)
        return;
    const rt_aim_seed sd = seeds[f];
    const int group =
        UNIFORM ? __builtin_amdgcn_readfirstlane(sd.group) : sd.group;
    const rt_surface *__restrict__ tabf = tab + (int64_t)group * nsurf;
    double zf;
    const int rc = rt_aim_chief(tabf, &sd, &args, fabs(sd.a0), &zf);
    const int axis = 1 - (m >> 1), sign = 1 - (m & 1);
    int rcm = 0;
    double val = NAN;
    if (!rc) {
        const double e = 2 * sign - 1.;
        double x;
        rcm = rt_aim_marginal(tabf, nsurf, &sd, &args, zf, axis == 0 ? e : 0.,
                              axis == 1 ? e : 0., &x);
        if (!rcm)
            val = e * fabs(x);
    }
    int st = rc;
    bool unreached = rc != 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int rk = __shfl(rcm, k, 4);
        if (!st && rk)
            st = rk;
        if (k < m && rk)
            unreached = true;
    }
    a[(f * 2 + sign) * 2 + axis] = unreached ? NAN : val;
    if (m == 0) {
        z[f] = zf;
        status[f] = st;
    }
}

/* ------------------------------------------------------------------ */
/* device-side consumers: rms, refocus sums, opd rays                 */
/* ------------------------------------------------------------------ */

#define RT_RED_BLOCKS 1024
#define RT_RED_THREADS 256

/* deterministic two-level sum of K accumulators: wave shuffle -> LDS ->
 * one partial per workgroup; the host adds the RT_RED_BLOCKS partials in
 * index order (no atomics, run-to-run identical) */
template <int K>
__device__ __forceinline__ void rt_block_reduce(double (&acc)[K],
                                                double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        for (int off = 32; off > 0; off >>= 1)
            acc[k] += __shfl_down(acc[k], off);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0)
#pragma unroll
        for (int k = 0; k < K; ++k)
            sm[wave][k] = acc[k];
    __syncthreads();
    if (threadIdx.x == 0)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            double v = sm[0][k];
            for (int w = 1; w < RT_RED_THREADS / 64; ++w)
                v += sm[w][k];
            partials[(int64_t)blockIdx.x * K + k] = v;
        }
}

/* second level on the device: one wavefront adds the per-workgroup partials
 * (lane l takes b = l, l + 64, ... in order, then a fixed shuffle tree), so
 * a two-pass consumer needs no host round trip between its passes */
__global__ void rt_finalize_kernel(const double *__restrict__ partials,
                                   int nblocks, int K,
                                   double *__restrict__ out)
{
    for (int k = 0; k < K; ++k) {
        double v = 0.;
        for (int b = threadIdx.x; b < nblocks; b += 64)
            v += partials[(int64_t)b * K + k];
        for (int off = 32; off > 0; off >>= 1)
            v += __shfl_down(v, off);
        if (threadIdx.x == 0)
            out[k] = v;
    }
}

/* sum of x and y of one row (rms: y.mean(0)) */
__global__ void rt_sum_xy_kernel(const double *__restrict__ Yrow, int64_t n,
                                 int64_t ld, double *__restrict__ partials)
{
    double acc[2] = {0., 0.};
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        acc[0] += Yrow[j];
        acc[1] += Yrow[ld + j];
    }
    rt_block_reduce<2>(acc, partials);
}

/* sum_k w_k ((x-x0)^2 + (y-y0)^2) */
__global__ void rt_rms_kernel(const double *__restrict__ Yrow,
                              const double *__restrict__ w, double wconst,
                              const double *__restrict__ sums, int64_t ref,
                              int64_t n, int64_t ld,
                              double *__restrict__ partials)
{
    /* centre: ray `ref`, or the plain mean from the sums of pass A */
    const double x0 = ref >= 0 ? Yrow[ref] : sums[0] / (double)n;
    const double y0 = ref >= 0 ? Yrow[ld + ref] : sums[1] / (double)n;
    double acc[1] = {0.};
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const double dx = Yrow[j] - x0, dy = Yrow[ld + j] - y0;
        const double r = dx * dx + dy * dy;
        acc[0] += r * (w ? w[j] : wconst);
    }
    rt_block_reduce<1>(acc, partials);
}

/* refocus pass A: over rays with finite u = i_xy/i_z: count, sum y, sum u */
__global__ void rt_refocus_sums_kernel(const double *__restrict__ Yrow,
                                       const double *__restrict__ Irow,
                                       int64_t n, int64_t ld,
                                       double *__restrict__ partials)
{
    double acc[5] = {0., 0., 0., 0., 0.};
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const double iz = Irow[2 * ld + j];
        const double ux = Irow[j] / iz, uy = Irow[ld + j] / iz;
        if (isfinite(ux) && isfinite(uy)) {
            acc[0] += 1.;
            acc[1] += Yrow[j];
            acc[2] += Yrow[ld + j];
            acc[3] += ux;
            acc[4] += uy;
        }
    }
    rt_block_reduce<5>(acc, partials);
}

/* refocus pass B: <w yc, uc> and <w uc, uc> with centred y, u */
__global__ void rt_refocus_dots_kernel(const double *__restrict__ Yrow,
                                       const double *__restrict__ Irow,
                                       const double *__restrict__ w,
                                       double wconst,
                                       const double *__restrict__ sums,
                                       int64_t n, int64_t ld,
                                       double *__restrict__ partials)
{
    /* means over the finite rays from the sums of pass A */
    const double my0 = sums[1] / sums[0], my1 = sums[2] / sums[0];
    const double mu0 = sums[3] / sums[0], mu1 = sums[4] / sums[0];
    double acc[2] = {0., 0.};
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const double iz = Irow[2 * ld + j];
        const double ux = Irow[j] / iz, uy = Irow[ld + j] / iz;
        if (isfinite(ux) && isfinite(uy)) {
            const double wk = w ? w[j] : wconst;
            const double y0 = Yrow[j] - my0, y1 = Yrow[ld + j] - my1;
            const double u0 = ux - mu0, u1 = uy - mu1;
            acc[0] += (wk * y0) * u0 + (wk * y1) * u1;
            acc[1] += (wk * u0) * u0 + (wk * u1) * u1;
        }
    }
    rt_block_reduce<2>(acc, partials);
}

/* max over rays of x^2 + y^2 of one row; NaN if any ray is NaN (np.max) */
__global__ void rt_r2max_kernel(const double *__restrict__ Yrow, int64_t n,
                                int64_t ld, double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][2];
    double mx = 0., bad = 0.;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n;
         j += (int64_t)gridDim.x * blockDim.x) {
        const double x = Yrow[j], y = Yrow[ld + j];
        const double r2 = x * x + y * y;
        if (r2 != r2)
            bad = 1.;
        else
            mx = r2 > mx ? r2 : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off), b = __shfl_down(bad, off);
        mx = o > mx ? o : mx;
        bad = b > bad ? b : bad;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wave][0] = mx;
        sm[wave][1] = bad;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < RT_RED_THREADS / 64; ++w) {
            mx = sm[w][0] > mx ? sm[w][0] : mx;
            bad = sm[w][1] > bad ? sm[w][1] : bad;
        }
        partials[(int64_t)blockIdx.x * 2] = mx;
        partials[(int64_t)blockIdx.x * 2 + 1] = bad;
    }
}

/*
 * Per-group spot statistics: the batch is `gridDim.y` contiguous groups of
 * group_rays rays (field x wavelength bundles as rt_generate_rays lays them
 * out); what GeometricTrace.rms() gives when called once per bundle
 * (geometric_trace.py:171-183), for every bundle in two passes over the row.
 * Rays whose intercept is not finite are left out and counted.
 * stats[g] = {count, mean x, mean y, sum w d^2 / sum w, max d^2, sum w}.
 */
#define RT_GRP_STATS 6

/* pass A: count, sum x, sum y, sum w over the finite rays of group g */
__global__ void rt_group_sums_kernel(const double *__restrict__ Yrow,
                                     const double *__restrict__ w,
                                     int64_t group_rays, int64_t ld,
                                     double *__restrict__ partials)
{
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    double acc[4] = {0., 0., 0., 0.};
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
         j < group_rays; j += (int64_t)gridDim.x * blockDim.x) {
        const double x = Yrow[base + j], y = Yrow[ld + base + j];
        if (isfinite(x) && isfinite(y)) {
            acc[0] += 1.;
            acc[1] += x;
            acc[2] += y;
            acc[3] += w ? w[base + j] : 1.;
        }
    }
    rt_block_reduce<4>(acc, partials + (int64_t)blockIdx.y * gridDim.x * 4);
}

/* one thread per group adds its pb partials in index order */
__global__ void rt_group_centroid_kernel(const double *__restrict__ partials,
                                         int pb, int ngroups,
                                         double *__restrict__ stats)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups)
        return;
    double a[4] = {0., 0., 0., 0.};
    for (int b = 0; b < pb; ++b)
        for (int k = 0; k < 4; ++k)
            a[k] += partials[((int64_t)g * pb + b) * 4 + k];
    double *s = stats + (int64_t)g * RT_GRP_STATS;
    s[0] = a[0];
    s[1] = a[1] / a[0];
    s[2] = a[2] / a[0];
    s[5] = a[3];
}

/* pass B: sum w d^2 and max d^2 about the centroid of group g */
__global__ void rt_group_spread_kernel(const double *__restrict__ Yrow,
                                       const double *__restrict__ w,
                                       int64_t group_rays, int64_t ld,
                                       const double *__restrict__ stats,
                                       double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][2];
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    const double x0 = stats[(int64_t)blockIdx.y * RT_GRP_STATS + 1];
    const double y0 = stats[(int64_t)blockIdx.y * RT_GRP_STATS + 2];
    double sum = 0., mx = 0.;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
         j < group_rays; j += (int64_t)gridDim.x * blockDim.x) {
        const double x = Yrow[base + j], y = Yrow[ld + base + j];
        if (isfinite(x) && isfinite(y)) {
            const double dx = x - x0, dy = y - y0;
            const double r = dx * dx + dy * dy;
            sum += r * (w ? w[base + j] : 1.);
            mx = r > mx ? r : mx;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off);
        const double o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wave][0] = sum;
        sm[wave][1] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int v = 1; v < RT_RED_THREADS / 64; ++v) {
            sum += sm[v][0];
            mx = sm[v][1] > mx ? sm[v][1] : mx;
        }
        double *p = partials + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        p[0] = sum;
        p[1] = mx;
    }
}

__global__ void rt_group_finish_kernel(const double *__restrict__ partials,
                                       int pb, int ngroups,
                                       double *__restrict__ stats)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups)
        return;
    double sum = 0., mx = 0.;
    for (int b = 0; b < pb; ++b) {
        const double *p = partials + ((int64_t)g * pb + b) * 2;
        sum += p[0];
        mx = p[1] > mx ? p[1] : mx;
    }
    double *s = stats + (int64_t)g * RT_GRP_STATS;
    s[3] = sum / s[5];
    s[4] = s[0] > 0. ? mx : __builtin_nan("");
}

/* reference-ray columns the opd kernel needs, all wave-uniform */
struct rt_opd_ref {
    double t[RT_MAX_SURFACES]; /* T[row][ref] */
    double y0[3], u0[3];       /* Y[0][ref], U[0][ref] */
    double ya[3], ua[3];       /* Y[after][ref], U[after][ref] */
    double yi[3];              /* Y[image][ref] */
};

/* transform + reference-sphere intercept of one ray (opd, :118-131) */
__device__ __forceinline__ void rt_opd_point(const rt_opd_args &a,
                                             const double (&yi_ref)[3],
                                             double (&y)[3], double (&u)[3],
                                             double &ti, double (&py)[3])
{
    if (a.rot_after) { /* ea.from_normal */
        rt_rot_from(a.r_after, y);
        rt_rot_from(a.r_after, u);
    }
    y[0] = y[0] + a.shift[0];
    y[1] = y[1] + a.shift[1];
    y[2] = y[2] + a.shift[2];
    if (a.rot_image) { /* ei.to_normal */
        rt_rot_to(a.r_image, y);
        rt_rot_to(a.r_image, u);
    }
    y[0] -= yi_ref[0];
    y[1] -= yi_ref[1];
    y[2] -= yi_ref[2];
    y[2] += a.radius;
    /* Spheroid(curvature=1/radius).intercept(y, u), elements.py:477-501 */
    const double c = 1. / a.radius;
    if (c == 0.) {
        ti = -y[2] / u[2];
    } else {
        const double uy = (u[0] * y[0] + u[1] * y[1]) + u[2] * y[2];
        const double yy = (y[0] * y[0] + y[1] * y[1]) + y[2] * y[2];
        const double d = c * uy - u[2];
        const double e = c * 1.;
        const double f = c * yy - 2. * y[2];
        const double g = sqrt(d * d - e * f);
        ti = -(d + g) / e;
    }
    py[0] = y[0] + ti * u[0];
    py[1] = y[1] + ti * u[1];
    py[2] = y[2] + ti * u[2];
    py[2] -= a.radius;
}

__global__ void rt_opd_kernel(rt_opd_args a, const rt_opd_ref *__restrict__ ref,
                              const double *__restrict__ Y,
                              const double *__restrict__ U,
                              const double *__restrict__ T, int64_t n,
                              int64_t ld, double *__restrict__ out)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n)
        return;
    /* t = (t[:after+1] - t[:after+1, ref]).sum(0): row by row */
    double t = 0.;
    for (int s = 0; s < a.nrows; ++s) {
        const double d = T[(int64_t)s * ld + j] - ref->t[s];
        t = s ? t + d : d;
    }
    if (!a.finite) { /* input reference sphere is a tilted plane (:104-109) */
        const double tj =
            (ref->u0[0] * (ref->y0[0] - Y[j]) +
             ref->u0[1] * (ref->y0[1] - Y[ld + j])) +
            ref->u0[2] * (ref->y0[2] - Y[2 * ld + j]);
        t -= tj * a.n0;
    }
    double y[3], u[3], py[3], ti;
    const int64_t ra = (int64_t)a.after * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        y[c] = Y[(ra + c) * ld + j];
        u[c] = U[(ra + c) * ld + j];
    }
    rt_opd_point(a, ref->yi, y, u, ti, py);
    /* the same for the reference ray (uniform; every lane recomputes it) */
    double yr[3], ur[3], pr[3], tr;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        yr[c] = ref->ya[c];
        ur[c] = ref->ua[c];
    }
    rt_opd_point(a, ref->yi, yr, ur, tr, pr);
    t += (ti - tr) * a.n_after;
    t = -t / a.lscale;
    out[j] = py[0] - pr[0];
    out[n + j] = py[1] - pr[1];
    out[2 * n + j] = t;
}


#endif /* RT_KERNELS_H */
