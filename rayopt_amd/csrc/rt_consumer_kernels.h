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

#endif /* RT_CONSUMER_KERNELS_H */
