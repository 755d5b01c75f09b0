/*
 * rt_consumer_kernels.h -- device code of what runs AFTER (or around) a trace
 * on device-resident rows: the aiming kernel (System.pupil for many fields),
 * the reductions behind rms / refocus / resize / spot statistics, and the
 * per-ray part of opd.  Included by rt_consumers.hip.
 */
#ifndef RT_CONSUMER_KERNELS_H
#define RT_CONSUMER_KERNELS_H

#include <hip/hip_runtime.h>
#include "rt_math.h"
#include "rt_aim.h"

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

/* a finishing kernel's last word: its results are in pinned memory, the
 * sequence number of the call follows them behind a system-scope fence, and
 * the host spins on that instead of synchronising the stream (a
 * hipStreamSynchronize costs about what a 30 us reduction's launch does) */
__device__ __forceinline__ void rt_sign(unsigned long long *ticket,
                                        unsigned long long seq)
{
    if (ticket) {
        __threadfence_system();
        __hip_atomic_store(ticket, seq, __ATOMIC_RELEASE,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

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

/*
 * A thread's share of rays lo..hi of C rows, as 16-byte loads, two per row in
 * flight.  A reduction is a pure read stream: this chip wants ~16 MB in
 * flight (8 TB/s x ~2 us), and 8-byte loads in a grid-stride loop had a
 * quarter of that (rms at 10^7 rays: 3.7 TB/s).  `tid` of `nthreads` is the
 * thread's place among those that share the range; the ray -> thread map is
 * fixed, so sums are run-to-run identical.
 */
template <int C, class F>
__device__ __forceinline__ void rt_stream_rows(const double *const (&row)[C],
                                               int64_t lo, int64_t hi,
                                               int64_t tid, int64_t nthreads,
                                               F &&use)
{
    double v[C];
    if ((lo & 1) && lo < hi) { /* rows are 16-byte aligned at even rays */
        if (tid == 0) {
#pragma unroll
            for (int c = 0; c < C; ++c)
                v[c] = row[c][lo];
            use(v);
        }
        ++lo;
    }
    const int64_t n2 = hi > lo ? (hi - lo) >> 1 : 0;
    int64_t p = tid;
    for (; p + nthreads < n2; p += 2 * nthreads) {
        double2 a[C], b[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            a[c] = ((const double2 *)(row[c] + lo))[p];
            b[c] = ((const double2 *)(row[c] + lo))[p + nthreads];
        }
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = a[c].x;
        use(v);
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = a[c].y;
        use(v);
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = b[c].x;
        use(v);
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = b[c].y;
        use(v);
    }
    if (p < n2) {
        double2 a[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            a[c] = ((const double2 *)(row[c] + lo))[p];
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = a[c].x;
        use(v);
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = a[c].y;
        use(v);
    }
    if (hi > lo && ((hi - lo) & 1) && tid == 0) {
#pragma unroll
        for (int c = 0; c < C; ++c)
            v[c] = row[c][hi - 1];
        use(v);
    }
}

#define RT_RED_TID ((int64_t)blockIdx.x * blockDim.x + threadIdx.x)
#define RT_RED_NTHREADS ((int64_t)gridDim.x * blockDim.x)

/* how the rays of a row lie in memory (rt_lay.h): ld doubles between the
 * components of a row (= rays per block), ts doubles from one block of the
 * batch to the next (0: one block, the plain layout) */
struct rt_pitch {
    int64_t ld, ts;
};
#define RT_AT(p, j) rt_block_col((p).ld, (p).ts, (j))

/* rt_stream_rows over rays lo..hi of a batch in blocks: block by block, each
 * a contiguous range.  ts[c] = doubles from block to block of row c (the
 * weights are one plain array: p.ld) */
template <int C, class F>
__device__ __forceinline__ void rt_stream_blocks(const double *const (&row)[C],
                                                 const int64_t (&ts)[C],
                                                 rt_pitch p, int64_t lo,
                                                 int64_t hi, int64_t tid,
                                                 int64_t nthreads, F &use)
{
    if (!p.ts) {
        rt_stream_rows(row, lo, hi, tid, nthreads, use);
        return;
    }
    for (int64_t b = lo / p.ld; b * p.ld < hi; ++b) {
        const int64_t first = b * p.ld;
        const double *rb[C];
#pragma unroll
        for (int c = 0; c < C; ++c)
            rb[c] = row[c] + b * ts[c];
        const double *const(&rows)[C] = rb;
        rt_stream_rows(rows, (lo > first ? lo : first) - first,
                       (hi < first + p.ld ? hi : first + p.ld) - first, tid,
                       nthreads, use);
    }
}

/* one wavefront adds the K columns of the per-workgroup partials side by
 * side (per column in the order of rt_finalize_kernel); every lane returns
 * with the sums */
template <int K>
__device__ __forceinline__ void rt_partial_sums(const double *__restrict__ p,
                                                int nblocks, double (&s)[K])
{
#pragma unroll
    for (int k = 0; k < K; ++k)
        s[k] = 0.;
    for (int b = threadIdx.x; b < nblocks; b += 64)
#pragma unroll
        for (int k = 0; k < K; ++k)
            s[k] += p[(int64_t)b * K + k];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        for (int off = 32; off > 0; off >>= 1)
            s[k] += __shfl_down(s[k], off);
        s[k] = __shfl(s[k], 0);
    }
}

/*
 * rms about the mean in ONE pass: sums of the coordinates shifted by ray 0
 * (d = y - y[0]), then  sum w |d - m|^2 = sum w d^2 - 2 m . sum w d +
 * |m|^2 sum w  with m = sum d / n.  Shifted by a ray of the bundle the
 * subtraction costs a few bits at most (|m| ~ spot size); how many is
 * reported (out[1] = sum w d^2 before the subtraction) and the caller falls
 * back to the two passes below when it is more than six.  With ref >= 0 the
 * centre is that ray and nothing is subtracted.  W: per-ray weights.
 * acc: sum dx, sum dy, sum w d^2 [, sum w dx, sum w dy, sum w].
 */
template <bool W>
__global__ void rt_rms_shifted_kernel(const double *__restrict__ Yrow,
                                      const double *__restrict__ w,
                                      int64_t ref, int64_t n, rt_pitch p,
                                      double *__restrict__ partials)
{
    const int64_t ld = p.ld, k0 = RT_AT(p, ref >= 0 ? ref : 0);
    const double x0 = Yrow[k0], y0 = Yrow[ld + k0];
    constexpr int K = W ? 6 : 3;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        acc[k] = 0.;
    if constexpr (W) {
        const double *const rows[3] = {Yrow, Yrow + ld, w};
        const int64_t ts[3] = {p.ts, p.ts, ld};
        auto use = [&](const double(&v)[3]) {
            const double dx = v[0] - x0, dy = v[1] - y0;
            const double r = dx * dx + dy * dy;
            acc[0] += dx;
            acc[1] += dy;
            acc[2] += r * v[2];
            acc[3] += v[2] * dx;
            acc[4] += v[2] * dy;
            acc[5] += v[2];
        };
        rt_stream_blocks(rows, ts, p, 0, n, RT_RED_TID, RT_RED_NTHREADS, use);
    } else {
        const double *const rows[2] = {Yrow, Yrow + ld};
        const int64_t ts[2] = {p.ts, p.ts};
        auto use = [&](const double(&v)[2]) {
            const double dx = v[0] - x0, dy = v[1] - y0;
            acc[0] += dx;
            acc[1] += dy;
            acc[2] += dx * dx + dy * dy;
        };
        rt_stream_blocks(rows, ts, p, 0, n, RT_RED_TID, RT_RED_NTHREADS, use);
    }
    rt_block_reduce<K>(acc, partials);
}

/* out[0] = sum w |y - centre|^2, out[1] = the same about ray 0 (what the
 * subtraction started from); out may be pinned host memory */
template <bool W>
__global__ void rt_rms_finish_kernel(const double *__restrict__ partials,
                                     int nblocks, int centred, double n,
                                     double *__restrict__ out,
                                     unsigned long long *ticket,
                                     unsigned long long seq)
{
    double s[W ? 6 : 3];
    rt_partial_sums(partials, nblocks, s);
    double A = s[2], swx, swy, sw;
    if constexpr (W) {
        swx = s[3];
        swy = s[4];
        sw = s[5];
    } else { /* w = 1/n for every ray */
        A /= n;
        swx = s[0] / n;
        swy = s[1] / n;
        sw = 1.;
    }
    if (threadIdx.x == 0) {
        const double mx = centred ? s[0] / n : 0., my = centred ? s[1] / n : 0.;
        out[0] = A - 2. * (mx * swx + my * swy) + (mx * mx + my * my) * sw;
        out[1] = A;
        rt_sign(ticket, seq);
    }
}

/* sum of x and y of one row (rms: y.mean(0)) */
__global__ void rt_sum_xy_kernel(const double *__restrict__ Yrow, int64_t n,
                                 rt_pitch p, double *__restrict__ partials)
{
    const int64_t ld = p.ld;
    double acc[2] = {0., 0.};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = RT_AT(p, k);
        acc[0] += Yrow[j];
        acc[1] += Yrow[ld + j];
    }
    rt_block_reduce<2>(acc, partials);
}

/* sum_k w_k ((x-x0)^2 + (y-y0)^2) */
__global__ void rt_rms_kernel(const double *__restrict__ Yrow,
                              const double *__restrict__ w, double wconst,
                              const double *__restrict__ sums, int64_t ref,
                              int64_t n, rt_pitch p,
                              double *__restrict__ partials)
{
    /* centre: ray `ref`, or the plain mean from the sums of pass A */
    const int64_t ld = p.ld, kr = ref >= 0 ? RT_AT(p, ref) : 0;
    const double x0 = ref >= 0 ? Yrow[kr] : sums[0] / (double)n;
    const double y0 = ref >= 0 ? Yrow[ld + kr] : sums[1] / (double)n;
    double acc[1] = {0.};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = RT_AT(p, k);
        const double dx = Yrow[j] - x0, dy = Yrow[ld + j] - y0;
        const double r = dx * dx + dy * dy;
        acc[0] += r * (w ? w[k] : wconst);
    }
    rt_block_reduce<1>(acc, partials);
}

/*
 * The refocus sums in ONE pass, shifted by ray 0 (y' = y - y[0], u' = u -
 * u[0]): over rays with finite u = i_xy / i_z
 *   acc: count, sum y' (2), sum u' (2), sum w y'.u', sum w |u'|^2,
 *        sum w |y'|^2 [, sum w, sum w y' (2), sum w u' (2)]
 * and the centred dots follow by subtraction in rt_refocus_finish_kernel.
 */
template <bool W>
__global__ void rt_refocus_shifted_kernel(const double *__restrict__ Yrow,
                                          const double *__restrict__ Irow,
                                          const double *__restrict__ w,
                                          int64_t n, rt_pitch p,
                                          double *__restrict__ partials)
{
    const int64_t ld = p.ld;
    const double iz0 = Irow[2 * ld];
    const double ky0 = Yrow[0], ky1 = Yrow[ld];
    const double ku0 = Irow[0] / iz0, ku1 = Irow[ld] / iz0;
    constexpr int K = W ? 13 : 8;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k)
        acc[k] = 0.;
    auto use = [&](double y0, double y1, double i0, double i1, double iz,
                   double wk) {
        const double ux = i0 / iz, uy = i1 / iz;
        if (isfinite(ux) && isfinite(uy)) {
            const double a0 = y0 - ky0, a1 = y1 - ky1;
            const double b0 = ux - ku0, b1 = uy - ku1;
            acc[0] += 1.;
            acc[1] += a0;
            acc[2] += a1;
            acc[3] += b0;
            acc[4] += b1;
            if constexpr (W) {
                acc[5] += (wk * a0) * b0 + (wk * a1) * b1;
                acc[6] += (wk * b0) * b0 + (wk * b1) * b1;
                acc[7] += (wk * a0) * a0 + (wk * a1) * a1;
                acc[8] += wk;
                acc[9] += wk * a0;
                acc[10] += wk * a1;
                acc[11] += wk * b0;
                acc[12] += wk * b1;
            } else {
                acc[5] += a0 * b0 + a1 * b1;
                acc[6] += b0 * b0 + b1 * b1;
                acc[7] += a0 * a0 + a1 * a1;
            }
        }
    };
    if constexpr (W) {
        const double *const rows[6] = {Yrow,        Yrow + ld,     Irow,
                                       Irow + ld,   Irow + 2 * ld, w};
        const int64_t ts[6] = {p.ts, p.ts, p.ts, p.ts, p.ts, ld};
        auto each = [&](const double(&v)[6]) {
            use(v[0], v[1], v[2], v[3], v[4], v[5]);
        };
        rt_stream_blocks(rows, ts, p, 0, n, RT_RED_TID, RT_RED_NTHREADS,
                         each);
    } else {
        const double *const rows[5] = {Yrow, Yrow + ld, Irow, Irow + ld,
                                       Irow + 2 * ld};
        const int64_t ts[5] = {p.ts, p.ts, p.ts, p.ts, p.ts};
        auto each = [&](const double(&v)[5]) {
            use(v[0], v[1], v[2], v[3], v[4], 1.);
        };
        rt_stream_blocks(rows, ts, p, 0, n, RT_RED_TID, RT_RED_NTHREADS,
                         each);
    }
    rt_block_reduce<K>(acc, partials);
}

/* out[0] = <w yc, uc>, out[1] = <w uc, uc>, out[2] = <w yc, yc> (c: centred
 * on the means over the finite rays), out[3] / out[4] = what the
 * subtractions for [1] / [2] started from */
template <bool W>
__global__ void rt_refocus_finish_kernel(const double *__restrict__ partials,
                                         int nblocks, double *__restrict__ out,
                                         unsigned long long *ticket,
                                         unsigned long long seq)
{
    double s[13];
    if constexpr (W) {
        rt_partial_sums(partials, nblocks, s);
    } else { /* w = 1 for every ray */
        double t[8];
        rt_partial_sums(partials, nblocks, t);
        for (int k = 0; k < 8; ++k)
            s[k] = t[k];
        s[8] = t[0];
        s[9] = t[1];
        s[10] = t[2];
        s[11] = t[3];
        s[12] = t[4];
    }
    if (threadIdx.x == 0) {
        const double my0 = s[1] / s[0], my1 = s[2] / s[0];
        const double mu0 = s[3] / s[0], mu1 = s[4] / s[0];
        out[0] = s[5] - (mu0 * s[9] + mu1 * s[10]) -
                 (my0 * s[11] + my1 * s[12]) + (my0 * mu0 + my1 * mu1) * s[8];
        out[1] = s[6] - 2. * (mu0 * s[11] + mu1 * s[12]) +
                 (mu0 * mu0 + mu1 * mu1) * s[8];
        out[2] = s[7] - 2. * (my0 * s[9] + my1 * s[10]) +
                 (my0 * my0 + my1 * my1) * s[8];
        out[3] = s[6];
        out[4] = s[7];
        rt_sign(ticket, seq);
    }
}

/* refocus pass A: over rays with finite u = i_xy/i_z: count, sum y, sum u */
__global__ void rt_refocus_sums_kernel(const double *__restrict__ Yrow,
                                       const double *__restrict__ Irow,
                                       int64_t n, rt_pitch p,
                                       double *__restrict__ partials)
{
    const int64_t ld = p.ld;
    double acc[5] = {0., 0., 0., 0., 0.};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = RT_AT(p, k);
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
                                       int64_t n, rt_pitch p,
                                       double *__restrict__ partials)
{
    /* means over the finite rays from the sums of pass A */
    const int64_t ld = p.ld;
    const double my0 = sums[1] / sums[0], my1 = sums[2] / sums[0];
    const double mu0 = sums[3] / sums[0], mu1 = sums[4] / sums[0];
    double acc[2] = {0., 0.};
    for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = RT_AT(p, k);
        const double iz = Irow[2 * ld + j];
        const double ux = Irow[j] / iz, uy = Irow[ld + j] / iz;
        if (isfinite(ux) && isfinite(uy)) {
            const double wk = w ? w[k] : wconst;
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
                                rt_pitch p, double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][2];
    double mx = 0., bad = 0.;
    const double *const rows[2] = {Yrow, Yrow + p.ld};
    const int64_t ts[2] = {p.ts, p.ts};
    auto use = [&](const double(&v)[2]) {
        const double r2 = v[0] * v[0] + v[1] * v[1];
        if (r2 != r2)
            bad = 1.;
        else
            mx = r2 > mx ? r2 : mx;
    };
    rt_stream_blocks(rows, ts, p, 0, n, RT_RED_TID, RT_RED_NTHREADS, use);
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

/* second level of rt_r2max_kernel: out[0] = max r^2, out[1] = 1 if any ray
 * was NaN; out may be pinned host memory */
__global__ void rt_r2max_finish_kernel(const double *__restrict__ partials,
                                       int nblocks, double *__restrict__ out,
                                       unsigned long long *ticket,
                                       unsigned long long seq)
{
    double mx = 0., bad = 0.;
    for (int b = threadIdx.x; b < nblocks; b += 64) {
        const double m = partials[(int64_t)b * 2], x = partials[(int64_t)b * 2 + 1];
        mx = m > mx ? m : mx;
        bad = x > bad ? x : bad;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off), b = __shfl_down(bad, off);
        mx = o > mx ? o : mx;
        bad = b > bad ? b : bad;
    }
    if (threadIdx.x == 0) {
        out[0] = mx;
        out[1] = bad;
        rt_sign(ticket, seq);
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
#define RT_GROUP_PINNED 4096 /* groups whose stats are written to the host */

/* pass A: count, sum x, sum y, sum w over the finite rays of group g */
__global__ void rt_group_sums_kernel(const double *__restrict__ Yrow,
                                     const double *__restrict__ w,
                                     int64_t group_rays, rt_pitch p,
                                     double *__restrict__ partials)
{
    const int64_t ld = p.ld;
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    double acc[4] = {0., 0., 0., 0.};
    auto use = [&](double x, double y, double wk) {
        if (isfinite(x) && isfinite(y)) {
            acc[0] += 1.;
            acc[1] += x;
            acc[2] += y;
            acc[3] += wk;
        }
    };
    if (w) {
        const double *const rows[3] = {Yrow, Yrow + ld, w};
        const int64_t ts[3] = {p.ts, p.ts, ld};
        auto each = [&](const double(&v)[3]) { use(v[0], v[1], v[2]); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
    } else {
        const double *const rows[2] = {Yrow, Yrow + ld};
        const int64_t ts[2] = {p.ts, p.ts};
        auto each = [&](const double(&v)[2]) { use(v[0], v[1], 1.); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
    }
    rt_block_reduce<4>(acc, partials + (int64_t)blockIdx.y * gridDim.x * 4);
}

/* one wavefront per group adds its pb partials (lane l takes b = l, l + 64,
 * ..., then a fixed shuffle tree) */
__global__ void rt_group_centroid_kernel(const double *__restrict__ partials,
                                         int pb, int ngroups,
                                         double *__restrict__ stats)
{
    const int g = blockIdx.x;
    double a[4];
    rt_partial_sums(partials + (int64_t)g * pb * 4, pb, a);
    if (threadIdx.x == 0) {
        double *s = stats + (int64_t)g * RT_GRP_STATS;
        s[0] = a[0];
        s[1] = a[1] / a[0];
        s[2] = a[2] / a[0];
        s[5] = a[3];
    }
}

/* pass B: sum w d^2 and max d^2 about the centroid of group g */
__global__ void rt_group_spread_kernel(const double *__restrict__ Yrow,
                                       const double *__restrict__ w,
                                       int64_t group_rays, rt_pitch p,
                                       const double *__restrict__ stats,
                                       double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][2];
    const int64_t ld = p.ld;
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    const double x0 = stats[(int64_t)blockIdx.y * RT_GRP_STATS + 1];
    const double y0 = stats[(int64_t)blockIdx.y * RT_GRP_STATS + 2];
    double sum = 0., mx = 0.;
    auto use = [&](double x, double y, double wk) {
        if (isfinite(x) && isfinite(y)) {
            const double dx = x - x0, dy = y - y0;
            const double r = dx * dx + dy * dy;
            sum += r * wk;
            mx = r > mx ? r : mx;
        }
    };
    if (w) {
        const double *const rows[3] = {Yrow, Yrow + ld, w};
        const int64_t ts[3] = {p.ts, p.ts, ld};
        auto each = [&](const double(&v)[3]) { use(v[0], v[1], v[2]); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
    } else {
        const double *const rows[2] = {Yrow, Yrow + ld};
        const int64_t ts[2] = {p.ts, p.ts};
        auto each = [&](const double(&v)[2]) { use(v[0], v[1], 1.); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
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

/* the finished stats go to `final` (the device array itself, or pinned host
 * memory) */
__global__ void rt_group_finish_kernel(const double *__restrict__ partials,
                                       int pb, int ngroups,
                                       const double *stats, double *final)
{
    const int g = blockIdx.x;
    double sum = 0., mx = 0.;
    for (int b = threadIdx.x; b < pb; b += 64) {
        const double *p = partials + ((int64_t)g * pb + b) * 2;
        sum += p[0];
        mx = p[1] > mx ? p[1] : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        sum += __shfl_down(sum, off);
        const double o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (threadIdx.x != 0)
        return;
    const double *s = stats + (int64_t)g * RT_GRP_STATS;
    double *f = final + (int64_t)g * RT_GRP_STATS;
    const double cnt = s[0], sw = s[5];
    f[0] = cnt;
    f[1] = s[1];
    f[2] = s[2];
    f[3] = sum / sw;
    f[4] = cnt > 0. ? mx : __builtin_nan("");
    f[5] = sw;
}

/*
 * rt_row_stats: what a caller asks of the image row -- per bundle the count of
 * surviving rays, the centroid, the rms about the centroid
 * (geometric_trace.py:171-183, ref=None), the rms about the bundle's reference
 * ray (:175, ref given; the spot diagrams of analysis.py:250-283 are centred
 * there) and the largest distance from the axis (resize, :185-193) -- in ONE
 * pass over x, y (and w) of the row: rt_rms, rt_spot_stats and rt_row_rmax read
 * the same 16 B per ray three times (four: the spread is a second pass) and
 * each pay a launch and a wait.  Sums are taken of coordinates shifted by a
 * ray of the bundle (the reference ray if it is given and finite, else the
 * first finite of the bundle's first 255 rays) and centred by subtraction in
 * rt_row_stats_finish_kernel, which also reports what the subtraction started
 * from (out[9]); more than six bits lost and the caller repeats with the two
 * passes.  acc: count, sum dx, sum dy, sum w, sum w dx, sum w dy, sum w d^2;
 * max r^2 about the axis beside them.
 */
#define RT_ROW_STATS 10
#define RT_ROW_ACC 8

template <bool W>
__global__ void rt_row_stats_kernel(const double *__restrict__ Yrow,
                                    const double *__restrict__ w,
                                    int64_t group_rays, int64_t ref, rt_pitch p,
                                    double *__restrict__ partials,
                                    double *__restrict__ shifts)
{
    __shared__ double sm[RT_RED_THREADS / 64][RT_ROW_ACC];
    __shared__ int first_wave[RT_RED_THREADS / 64];
    const int64_t ld = p.ld;
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    /* the shift: thread 0 looks at the reference ray (where there is one),
     * the others at the first rays of the bundle; the first finite one wins --
     * the same in every workgroup of the bundle */
    double cx, cy;
    {
        int64_t c = ref >= 0 ? (threadIdx.x ? threadIdx.x - 1 : ref)
                             : threadIdx.x;
        const bool in = c < group_rays;
        const int64_t j = RT_AT(p, base + (in ? c : 0));
        cx = Yrow[j];
        cy = Yrow[ld + j];
        const unsigned long long ok =
            __ballot(in && isfinite(cx) && isfinite(cy));
        if (lane == 0)
            first_wave[wave] = ok ? __ffsll((long long)ok) - 1 : -1;
    }
    __syncthreads();
    int src = -1;
    for (int v = RT_RED_THREADS / 64 - 1; v >= 0; --v)
        if (first_wave[v] >= 0)
            src = v * 64 + first_wave[v];
    /* (broadcast through LDS: the winner writes, everybody reads) */
    __shared__ double sxy[2];
    if ((int)threadIdx.x == src) {
        sxy[0] = cx;
        sxy[1] = cy;
    }
    __syncthreads();
    const double sx = src >= 0 ? sxy[0] : 0., sy = src >= 0 ? sxy[1] : 0.;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        double *sh = shifts + (int64_t)blockIdx.y * 4;
        sh[0] = sx;
        sh[1] = sy;
        sh[2] = ref >= 0 ? cx : __builtin_nan(""); /* thread 0 holds the */
        sh[3] = ref >= 0 ? cy : __builtin_nan(""); /* reference ray      */
    }
    double acc[7] = {0., 0., 0., 0., 0., 0., 0.}, mx = 0.;
    auto use = [&](double x, double y, double wk) {
        if (isfinite(x) && isfinite(y)) {
            const double dx = x - sx, dy = y - sy;
            const double r2 = x * x + y * y;
            acc[0] += 1.;
            acc[1] += dx;
            acc[2] += dy;
            acc[3] += wk;
            acc[4] += wk * dx;
            acc[5] += wk * dy;
            acc[6] += wk * (dx * dx + dy * dy);
            mx = r2 > mx ? r2 : mx;
        }
    };
    if constexpr (W) {
        const double *const rows[3] = {Yrow, Yrow + ld, w};
        const int64_t ts[3] = {p.ts, p.ts, ld};
        auto each = [&](const double(&v)[3]) { use(v[0], v[1], v[2]); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
    } else {
        const double *const rows[2] = {Yrow, Yrow + ld};
        const int64_t ts[2] = {p.ts, p.ts};
        auto each = [&](const double(&v)[2]) { use(v[0], v[1], 1.); };
        rt_stream_blocks(rows, ts, p, base, base + group_rays, RT_RED_TID,
                         RT_RED_NTHREADS, each);
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        for (int off = 32; off > 0; off >>= 1)
            acc[k] += __shfl_down(acc[k], off);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k)
            sm[wave][k] = acc[k];
        sm[wave][7] = mx;
    }
    __syncthreads();
    if (threadIdx.x < RT_ROW_ACC) {
        const int k = threadIdx.x;
        double v = sm[0][k];
        for (int q = 1; q < RT_RED_THREADS / 64; ++q)
            v = k == 7 ? (sm[q][k] > v ? sm[q][k] : v) : v + sm[q][k];
        partials[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * RT_ROW_ACC +
                 k] = v;
    }
}

/*
 * One wavefront per bundle: the pb partials in index order, then
 * final[g] = {count, sum w, mean x, mean y, sum w |y - mean|^2 / sum w,
 *             sum w |y - y_ref|^2 / sum w (NaN without a reference ray),
 *             max (x^2 + y^2) (NaN for an empty bundle), weighted centroid x, y,
 *             sum w |y - shift|^2 / sum w (what the subtractions started from)}
 * followed -- `ticket` not NULL -- by the sequence number of the call, after
 * a system-scope fence: the host spins on it instead of waiting for the
 * stream.
 */
__global__ void rt_row_stats_finish_kernel(const double *__restrict__ partials,
                                           int pb, int ngroups,
                                           const double *__restrict__ shifts,
                                           double *final,
                                           unsigned long long *ticket,
                                           unsigned long long seq,
                                           unsigned int *arrived)
{
    const int g = blockIdx.x;
    double s[7] = {0., 0., 0., 0., 0., 0., 0.}, mx = 0.;
    for (int b = threadIdx.x; b < pb; b += 64) {
        const double *q = partials + ((int64_t)g * pb + b) * RT_ROW_ACC;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            s[k] += q[k];
        mx = q[7] > mx ? q[7] : mx;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k)
        for (int off = 32; off > 0; off >>= 1)
            s[k] += __shfl_down(s[k], off);
    for (int off = 32; off > 0; off >>= 1) {
        const double o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (threadIdx.x != 0)
        return;
    const double *sh = shifts + (int64_t)g * 4;
    double *f = final + (int64_t)g * RT_ROW_STATS;
    const double cnt = s[0], sw = s[3], A = s[6];
    const double mx_ = s[1] / cnt, my_ = s[2] / cnt;
    const double rx = sh[2] - sh[0], ry = sh[3] - sh[1];
    f[0] = cnt;
    f[1] = sw;
    f[2] = sh[0] + mx_;
    f[3] = sh[1] + my_;
    f[4] = (A - 2. * (mx_ * s[4] + my_ * s[5]) + (mx_ * mx_ + my_ * my_) * sw) /
           sw;
    f[5] = (A - 2. * (rx * s[4] + ry * s[5]) + (rx * rx + ry * ry) * sw) / sw;
    f[6] = cnt > 0. ? mx : __builtin_nan("");
    f[7] = sh[0] + s[4] / sw;
    f[8] = sh[1] + s[5] / sw;
    f[9] = A / sw;
    if (ticket) {
        /* the last bundle to finish signs for all of them */
        __threadfence_system();
        if (atomicAdd(arrived, 1u) == (unsigned)ngroups - 1) {
            *arrived = 0;
            __threadfence_system();
            __hip_atomic_store(ticket, seq, __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
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

/*
 * The reference-ray columns of every bundle, gathered on the device (one
 * thread per bundle): refs[g] = the rows of ray g * group_rays + a.ref.
 */
__global__ void rt_opd_refs_kernel(rt_opd_args a, const double *__restrict__ Y,
                                   const double *__restrict__ U,
                                   const double *__restrict__ T,
                                   int64_t group_rays, int ngroups, rt_pitch p,
                                   rt_opd_ref *__restrict__ refs)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= ngroups)
        return;
    const int64_t ld = p.ld, j = RT_AT(p, (int64_t)g * group_rays + a.ref);
    rt_opd_ref *r = refs + g;
    for (int s = 0; s < a.nrows; ++s)
        r->t[s] = T[(int64_t)s * ld + j];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        r->y0[c] = Y[(int64_t)c * ld + j];
        r->u0[c] = U[(int64_t)c * ld + j];
        r->ya[c] = Y[((int64_t)a.after * 3 + c) * ld + j];
        r->ua[c] = U[((int64_t)a.after * 3 + c) * ld + j];
        r->yi[c] = Y[((int64_t)a.image * 3 + c) * ld + j];
    }
}

/* x, y on the reference sphere and the path difference t (waves) of ray k
 * against the reference ray `ref` of its bundle: GeometricTrace.opd, :101-131,
 * operation for operation */
__device__ __forceinline__ void rt_opd_ray(const rt_opd_args &a,
                                           const rt_opd_ref *__restrict__ ref,
                                           const double *__restrict__ Y,
                                           const double *__restrict__ U,
                                           const double *__restrict__ T,
                                           int64_t ld, int64_t j, double &x,
                                           double &yy, double &t)
{
    /* t = (t[:after+1] - t[:after+1, ref]).sum(0): row by row */
    t = 0.;
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
    x = py[0] - pr[0];
    yy = py[1] - pr[1];
}

/*
 * rt_opd_stats, first level: the path difference of every ray of bundle
 * blockIdx.y and, over the rays where x, y and t are finite (what the
 * reference keeps before it resamples, :133-135), the sums
 *   count, sum w, sum w t, sum w t^2, min t, max t
 * -- t is measured from the bundle's reference ray (t_ref = 0), so the plain
 * sums lose nothing that matters at 1e-9.  keep: x | y | t of every ray go to
 * out[3][n] as well (rt_opd_device).  One ray per lane and pass: 18-21
 * independent 8-byte loads each, plenty in flight.
 */
#define RT_OPD_SUMS 6
__global__ void rt_opd_stats_kernel(rt_opd_args a,
                                    const rt_opd_ref *__restrict__ refs,
                                    const double *__restrict__ Y,
                                    const double *__restrict__ U,
                                    const double *__restrict__ T,
                                    const double *__restrict__ w,
                                    int64_t group_rays, int64_t n, rt_pitch p,
                                    double *__restrict__ out,
                                    double *__restrict__ partials)
{
    __shared__ double sm[RT_RED_THREADS / 64][RT_OPD_SUMS];
    const rt_opd_ref *ref = refs + blockIdx.y;
    const int64_t base = (int64_t)blockIdx.y * group_rays;
    double cnt = 0., sw = 0., s1 = 0., s2 = 0.;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int64_t q = RT_RED_TID; q < group_rays; q += RT_RED_NTHREADS) {
        const int64_t k = base + q;
        double x, y, t;
        rt_opd_ray(a, ref, Y, U, T, p.ld, RT_AT(p, k), x, y, t);
        if (out) {
            out[k] = x;
            out[n + k] = y;
            out[2 * n + k] = t;
        }
        if (isfinite(x) && isfinite(y) && isfinite(t)) {
            const double wk = w ? w[k] : 1.;
            cnt += 1.;
            sw += wk;
            s1 += wk * t;
            s2 += wk * t * t;
            lo = t < lo ? t : lo;
            hi = t > hi ? t : hi;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        sw += __shfl_down(sw, off);
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
        const double l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sm[wave][0] = cnt;
        sm[wave][1] = sw;
        sm[wave][2] = s1;
        sm[wave][3] = s2;
        sm[wave][4] = lo;
        sm[wave][5] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int v = 1; v < RT_RED_THREADS / 64; ++v) {
            cnt += sm[v][0];
            sw += sm[v][1];
            s1 += sm[v][2];
            s2 += sm[v][3];
            lo = sm[v][4] < lo ? sm[v][4] : lo;
            hi = sm[v][5] > hi ? sm[v][5] : hi;
        }
        double *q = partials +
                    ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * RT_OPD_SUMS;
        q[0] = cnt;
        q[1] = sw;
        q[2] = s1;
        q[3] = s2;
        q[4] = lo;
        q[5] = hi;
    }
}

/* second level, one wavefront per bundle: stats[g] = {count, sum w, mean,
 * rms about the mean, min, max, peak to valley, rms about the reference ray}
 * (RT_OPD_STATS doubles), into the device array or pinned host memory */
__global__ void rt_opd_finish_kernel(const double *__restrict__ partials,
                                     int pb, double *__restrict__ final)
{
    const int g = blockIdx.x;
    double cnt = 0., sw = 0., s1 = 0., s2 = 0.;
    double lo = __builtin_inf(), hi = -__builtin_inf();
    for (int b = threadIdx.x; b < pb; b += 64) {
        const double *q = partials + ((int64_t)g * pb + b) * RT_OPD_SUMS;
        cnt += q[0];
        sw += q[1];
        s1 += q[2];
        s2 += q[3];
        lo = q[4] < lo ? q[4] : lo;
        hi = q[5] > hi ? q[5] : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        cnt += __shfl_down(cnt, off);
        sw += __shfl_down(sw, off);
        s1 += __shfl_down(s1, off);
        s2 += __shfl_down(s2, off);
        const double l2 = __shfl_down(lo, off), h2 = __shfl_down(hi, off);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if (threadIdx.x != 0)
        return;
    double *f = final + (int64_t)g * RT_OPD_STATS;
    const double nan = __builtin_nan("");
    const double mean = s1 / sw, var = s2 / sw - mean * mean;
    f[0] = cnt;
    f[1] = sw;
    f[2] = cnt > 0. ? mean : nan;
    f[3] = cnt > 0. ? sqrt(var > 0. ? var : 0.) : nan;
    f[4] = cnt > 0. ? lo : nan;
    f[5] = cnt > 0. ? hi : nan;
    f[6] = cnt > 0. ? hi - lo : nan;
    f[7] = cnt > 0. ? sqrt(s2 / sw) : nan;
}

__global__ void rt_opd_kernel(rt_opd_args a, const rt_opd_ref *__restrict__ ref,
                              const double *__restrict__ Y,
                              const double *__restrict__ U,
                              const double *__restrict__ T, int64_t n,
                              rt_pitch p, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n)
        return;
    double x, y, t;
    rt_opd_ray(a, ref, Y, U, T, p.ld, RT_AT(p, k), x, y, t);
    out[k] = x;
    out[n + k] = y;
    out[2 * n + k] = t;
}

#endif /* RT_CONSUMER_KERNELS_H */
