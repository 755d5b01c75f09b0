/*
 * rt_probe_kernels.h -- LABORATORY device code, compiled only into
 * librt_mi355_probes.so (-DRT_BUILD_PROBES): the trace kernel's rejected
 * variants (2 / 4 rays per lane, non-temporal stores, XCD-contiguous dealing,
 * chip-wide read gating, fake-uniform input reads) and the bandwidth probes
 * (store pattern without arithmetic, fills, copy).  Everything here was
 * measured and decided (profiles/HISTORY.md); none of it is in the shipped
 * library.
 */
#ifndef RT_PROBE_KERNELS_H
#define RT_PROBE_KERNELS_H

#ifndef RT_BUILD_PROBES
#error "rt_probe_kernels.h is part of the laboratory build (-DRT_BUILD_PROBES)"
#endif

#include "rt_march.h"

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

template <int R, bool NT, bool XCD>
__global__ void rt_trace_lab_kernel(const rt_surface *__restrict__ surf, int start,
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

#endif /* RT_PROBE_KERNELS_H */
