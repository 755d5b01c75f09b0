/*
 * rt_math.h -- per-ray surface arithmetic of the sequential trace, FP64.
 *
 * One call of rt_step<R>() is one "ray-surface op" for R rays held in
 * registers: transfer into the element frame -> intercept (plane / closed
 * form sphere+conic / Newton for even aspheres) -> clip -> Snell refraction
 * or reflection.  It follows, operation for operation and in the same
 * association order, what the reference evaluates with numpy:
 *
 *   System.propagate            rayopt/system.py:459-464
 *   TransformMixin._do_rotate   rayopt/elements.py:156-175
 *   Interface.propagate         rayopt/elements.py:306-315
 *   Element.clip                rayopt/elements.py:206-209
 *   Spheroid.intercept          rayopt/elements.py:477-501
 *   Interface.intercept         rayopt/elements.py:333-349  (+ scipy newton)
 *   Spheroid.surface_sag        rayopt/elements.py:440-455
 *   Spheroid.surface_normal     rayopt/elements.py:457-475
 *   Interface.refract           rayopt/elements.py:351-369
 *
 * numpy never contracts a*b+c into an FMA and adds the three terms of a
 * length-3 .sum(1) left to right, so this file must be compiled with
 * -ffp-contract=off; sqrt and / are IEEE correctly rounded on gfx950 FP64
 * (and in numpy), which makes the spherical path reproduce the reference to
 * the last bit or two.
 *
 * Everything that depends only on the surface (flags, curvature, mu ...) is
 * wave-uniform: the kernel reads it through scalar loads into SGPRs and the
 * branches below are scalar branches, never divergence.  The only divergent
 * construct is the Newton loop, whose trip count is decided per wave with a
 * ballot (RT_WAVE_ANY).
 *
 * The header is also compiled for the host by tests/hostemu (g++), purely so
 * the arithmetic can be checked in a container without a GPU; the product
 * never runs it on the CPU.
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <math.h>
#include "../../include/rt_mi355.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
/* true if the predicate holds on any lane of the 64-wide wavefront */
#define RT_WAVE_ANY(p) (__ballot(p) != 0ull)
#define RT_NAN __builtin_nan("")
#else
#define RT_WAVE_ANY(p) (p)
#define RT_NAN __builtin_nan("")
#endif

/*
 * The 3x3 products are the one place where the reference leaves numpy's
 * elementwise arithmetic: np.dot(y, R) of an (N,3) array (elements.py:156-162)
 * is a BLAS dgemm, and a dgemm micro-kernel accumulates over the inner index
 * in order with fused multiply-adds -- out = fma(c, r2, fma(b, r1, a*r0)).
 * Checked entry by entry against exact rational arithmetic for numpy's
 * OpenBLAS (0.3.29, x86-64 with FMA3; N = 2 ... 10^5, R and R.T): every value
 * follows this chain, none the unfused sum.  The explicit fma here is part of
 * the numerical contract like -ffp-contract=off is everywhere else: with it
 * tilted systems are bit-identical to the reference's goldens too.
 */
RT_HD double rt_dot3(double a, double b, double c, double r0, double r1,
                     double r2)
{
    return __builtin_fma(c, r2, __builtin_fma(b, r1, a * r0));
}

/* y @ R.T : to_normal (elements.py:174-175), row vector times R transposed */
RT_HD void rt_rot_to(const double *__restrict__ r, double (&v)[3])
{
    const double a = v[0], b = v[1], c = v[2];
    v[0] = rt_dot3(a, b, c, r[0], r[1], r[2]);
    v[1] = rt_dot3(a, b, c, r[3], r[4], r[5]);
    v[2] = rt_dot3(a, b, c, r[6], r[7], r[8]);
}

/* y @ R : from_normal (elements.py:171-172) */
RT_HD void rt_rot_from(const double *__restrict__ r, double (&v)[3])
{
    const double a = v[0], b = v[1], c = v[2];
    v[0] = rt_dot3(a, b, c, r[0], r[3], r[6]);
    v[1] = rt_dot3(a, b, c, r[1], r[4], r[7]);
    v[2] = rt_dot3(a, b, c, r[2], r[5], r[8]);
}

/* ------------------------------------------------------------------ */
/* IEEE quotients and square roots without the range scaffolding      */
/* ------------------------------------------------------------------ */
/*
 * `a / b` and `sqrt(x)` in FP64 are not instructions on gfx950 but sequences
 * the compiler expands (checked in the ISA of this file's kernels):
 *
 *   a / b   v_div_scale x2, v_rcp, 4 FMA (two Newton steps on the reciprocal),
 *           v_mul, v_fma (residual), v_div_fmas, v_div_fixup      11 VALU
 *   sqrt    v_cmp + v_cndmask + v_ldexp (scale up below 2^-767), v_rsq,
 *           2 v_mul, 7 FMA, v_cndmask + v_ldexp (scale back),
 *           v_cmp_class + 2 v_cndmask (0 and inf pass through)    18 VALU
 *
 * The scaffolding -- div_scale / div_fmas / div_fixup, the ldexp pair --
 * only ACTS at the ends of the exponent range: v_div_scale returns its
 * operand unchanged and clears VCC unless an operand is zero / denormal /
 * infinite / NaN, the exponents differ by >= 768, or the quotient or the
 * reciprocal would be denormal; v_div_fmas with VCC clear IS v_fma;
 * v_div_fixup returns the quotient it is handed unless an operand is zero,
 * infinite or NaN.  So for operands with magnitudes in [2^-100, 2^100] the
 * eleven instructions compute exactly what the seven in rt_rcp_refined +
 * rt_quot compute -- the same instructions on the same operands -- and the
 * results are the same bits.  What that buys: (i) two quotients by the same
 * denominator share the reciprocal (refraction: a = mu (u.r) / r^2 and
 * b = (mu^2 - 1) / r^2, elements.py:358-366), (ii) a quotient by a
 * wave-uniform denominator (the sphere's -(d + g) / e with e = c,
 * elements.py:492-500) takes its refined reciprocal from the surface table
 * (rt_surface.rc, filled by the device when the table is uploaded).
 *
 * The range is CHECKED, per wavefront, with two or three compares: a
 * wavefront in which any lane holds a finite operand outside the range takes
 * the compiler's own sequence for all its lanes (scalar branch).  NaN lanes
 * (dead rays) do not count: both forms give NaN.  Zeros and infinities do
 * count (the fix-up cases).  RT_F_RANGE on the element says its wave-uniform
 * operands (c, mu^2 - 1) are inside the range; without it the plain form
 * runs.  rt_selftest_arith() (include/rt_mi355.h) compares both forms on the
 * device bit for bit, operands drawn across and beyond the range.
 *
 * The host build (tests/hostemu) has IEEE `/` and sqrt() and uses them.
 */
#define RT_RANGE_BIG 0x1p100
#define RT_RANGE_TINY 0x1p-100
#define RT_SQRT_SCALED_BELOW 0x1p-767

#if defined(__HIP_DEVICE_COMPILE__)
/* the reciprocal as the division sequence refines it: v_rcp + two steps */
RT_HD double rt_rcp_refined(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.), r);
    r = __builtin_fma(r, __builtin_fma(-d, r, 1.), r);
    return r;
}

/* n / d given r = rt_rcp_refined(d): v_mul, residual, v_div_fmas (VCC = 0) */
RT_HD double rt_quot(double n, double d, double r)
{
    const double q = n * r;
    return __builtin_fma(__builtin_fma(-d, q, n), r, q);
}

/* sqrt(x) for x outside (0, 2^-767): the sequence without its ldexp pair */
RT_HD double rt_sqrt_unscaled(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = y * .5;
    const double r = __builtin_fma(-h, g, .5);
    g = __builtin_fma(g, r, g);
    double d = __builtin_fma(-g, g, x);
    h = __builtin_fma(h, r, h);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    /* +-0 and +inf pass through (v_cmp_class 0x260) */
    return __builtin_isfpclass(x, 0x260) ? x : g;
}
#else
/* host build: IEEE division and sqrt as they are */
RT_HD double rt_rcp_refined(double d) { return 1. / d; }
RT_HD double rt_quot(double n, double d, double r) { (void)r; return n / d; }
RT_HD double rt_sqrt_unscaled(double x) { return sqrt(x); }
#endif
/* a finite magnitude outside [2^-100, 2^100] (zero and inf included)? */
#define RT_ODD_MAG(x)                                                         \
    (__builtin_fabs(x) >= RT_RANGE_BIG || __builtin_fabs(x) < RT_RANGE_TINY)

/*
 * sqrt(1 - (1+k) c^2 r^2): the one square root surface_sag (:451) and
 * surface_normal (:468) both evaluate at the same point; computed once per
 * point and handed to both (identical bits, half the sqrt sequences).
 */
RT_HD double rt_conic_root(const rt_surface *__restrict__ S, unsigned flags,
                           double r2)
{
    /* 1 - p is exact for p in [1/2, 2] (Sterbenz) and > 1/2 below: the
     * argument is <= 0, NaN, or >= 2^-53 -- never in (0, 2^-767) */
    return (flags & RT_F_CURVED) ? rt_sqrt_unscaled(1. - S->kc2 * r2) : 1.;
}

/*
 * The element's aspheric terms held in registers (wave-uniform: SGPRs) for
 * the length of a Newton solve.  Read from the table inside the iteration
 * they cost a scalar load and a wait each -- per term, per iterate (the ISA
 * of round 4's exact path: fourteen dependent scalar-load round trips per
 * iterate).  NA = the term-count class (4 / 7 / RT_MAX_ASPH); exactly `n`
 * terms are evaluated, in the reference's order: the same operations on the
 * same operands, the same bits.
 */
template <int NA> struct rt_asph_terms {
    double a[NA], da[NA];
    double c, kc2;
    int n;
};

template <int NA>
RT_HD void rt_asph_read(const rt_surface *__restrict__ S, unsigned flags,
                        rt_asph_terms<NA> &A)
{
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        A.a[i] = S->asph[i];
        A.da[i] = S->dasph[i];
    }
    A.n = (flags & RT_F_ASPH) ? (S->nasph < NA ? S->nasph : NA) : 0;
    A.c = S->c;
    A.kc2 = S->kc2;
}

/* Spheroid.surface_sag(p) residual, elements.py:440-455 */
template <int NA>
RT_HD double rt_sag_t(const rt_asph_terms<NA> &A, unsigned flags, double r2,
                      double pz, double root)
{
    double e = pz;
    if (flags & RT_F_CURVED)
        e -= (A.c * r2) / (1. + root);
    if (flags & RT_F_ASPH) {
        double d = 0.;
#pragma unroll
        for (int i = NA - 1; i >= 0; --i) {
            if (i < A.n) {
                d += A.a[i];
                d *= r2;
            }
        }
        e -= d;
    }
    return e;
}

/* x,y scale factor e of Spheroid.surface_normal, q = (x e, y e, 1), :457-475 */
template <int NA>
RT_HD double rt_normal_e_t(const rt_asph_terms<NA> &A, unsigned flags,
                           double r2, double root)
{
    double e = 0.;
    if (flags & RT_F_CURVED)
        e -= A.c / root;
    if (flags & RT_F_ASPH) {
        double d = 0.;
#pragma unroll
        for (int i = NA - 1; i >= 0; --i) {
            if (i < A.n) {
                d *= r2;
                d += A.da[i];
            }
        }
        e -= d;
    }
    return e;
}

/* the same with the terms read from the table as they are needed (one
 * evaluation per ray-surface op: the refraction's normal) */
RT_HD double rt_normal_e(const rt_surface *__restrict__ S, unsigned flags,
                         double r2, double root)
{
    double e = 0.;
    if (flags & RT_F_CURVED)
        e -= S->c / root;
    if (flags & RT_F_ASPH) {
        double d = 0.;
        for (int i = S->nasph - 1; i >= 0; --i) {
            d *= r2;
            d += S->dasph[i];
        }
        e -= d;
    }
    return e;
}

/* np.isclose(p, p0, rtol=0, atol=tol) as scipy's newton uses it */
RT_HD bool rt_isclose(double p, double p0, double tol)
{
    if (isfinite(p) && isfinite(p0))
        return fabs(p - p0) <= tol;
    return p == p0; /* equal infinities are close, NaN never is */
}

/*
 * Interface.intercept (elements.py:333-349): per-ray scipy.optimize.newton,
 * scalar Newton-Raphson branch (scipy/optimize/_zeros_py.py, 1.15.3) with
 * x0 = plane intercept, tol = 1e-7 (absolute, rtol = 0), maxiter = 5:
 *   fval == 0            -> return current iterate
 *   fder == 0            -> RuntimeError -> NaN
 *   p = p0 - fval/fder;  isclose(p, p0) -> return p
 *   5 iterations without convergence -> RuntimeError -> NaN
 * All R rays of the lane iterate together; the wave leaves the loop when no
 * lane of the wavefront has a ray still iterating (ballot).
 */
/* One iterate for one ray, every operation as the reference's: returns true
 * when the ray is finished (res = the root, or NaN left in place), false when
 * s has been updated and the iteration goes on. */
template <int NA>
RT_HD bool rt_newton_iterate_t(const rt_asph_terms<NA> &A, unsigned flags,
                               const double (&y)[3], const double (&u)[3],
                               double &s, double &res)
{
    const double px = y[0] + s * u[0];
    const double py = y[1] + s * u[1];
    const double pz = y[2] + s * u[2];
    const double r2 = px * px + py * py;
    /* sqrt(1 - (1+k) c^2 r^2), the one square root sag and normal share
     * (rt_conic_root) */
    const double root = (flags & RT_F_CURVED)
                            ? rt_sqrt_unscaled(1. - A.kc2 * r2) : 1.;
    const double fval = rt_sag_t<NA>(A, flags, r2, pz, root);
    if (fval == 0.) {
        res = s;
        return true;
    }
    const double e = rt_normal_e_t<NA>(A, flags, r2, root);
    /* np.dot(normal, u.T) of a (1,3) and a (3,1) array (:342): BLAS, i.e.
     * the fused chain of rt_dot3 */
    const double fder = rt_dot3(px * e, py * e, 1., u[0], u[1], u[2]);
    if (fder == 0.)
        return true; /* "Derivative was zero" -> NaN */
    const double p = s - fval / fder;
    if (rt_isclose(p, s, 1e-7)) {
        res = p;
        return true;
    }
    s = p;
    return false;
}

/* ... with the terms read from the table (the rim points of the default
 * arithmetic: a handful of rays) */
RT_HD bool rt_newton_iterate(const rt_surface *__restrict__ S, unsigned flags,
                             const double (&y)[3], const double (&u)[3],
                             double &s, double &res)
{
    rt_asph_terms<RT_MAX_ASPH> A;
    rt_asph_read<RT_MAX_ASPH>(S, flags, A);
    return rt_newton_iterate_t<RT_MAX_ASPH>(A, flags, y, u, s, res);
}

/*
 * CENSUS / `census` (measurement only: a template parameter, so that the
 * product kernels' instantiations carry no trace of it -- the code the
 * compiler makes of this file is touchy: round 6 saw 153 -> 238 VALU
 * instructions per ray-surface op in the generated-batch kernel, on a system
 * WITHOUT aspheres, from a counter behind a NULL check in here):
 * census[0] counts the trips of this lane's wavefront through the iteration,
 * census[1] the iterates this lane's own rays needed, census[2] the solves the
 * wavefront entered with a live ray -- the ratio of the first two over a
 * launch is how many of the lanes a wavefront drags through the loop were
 * still iterating (rt_newton_census).
 *
 * A ray whose iterate has become NaN is retired: NaN is never close to
 * anything and every later iterate is NaN again, so the reference's solver
 * runs out of iterations and yields NaN (elements.py:345-348) -- the result is
 * decided.  That is every ray that ARRIVES dead (its direction NaN-poisoned by
 * an earlier clip, miss or total reflection): after one trip instead of five,
 * so that one vignetted ray no longer holds its wavefront for the whole
 * iteration at every asphere behind the stop.  (Retiring such rays before
 * the loop, on their direction, does the same in zero trips and was measured
 * 3.8 % slower on C4, where no ray is dead -- the compiler's doing, not the
 * compare's; this form costs 0.7 %: scripts/variant_ab.py, round 6.)
 */
template <int R, int NA, bool CENSUS = false>
RT_HD void rt_newton(const rt_surface *__restrict__ S, unsigned flags,
                     const double (&y)[R][3], const double (&u)[R][3],
                     double (&s)[R], unsigned *census = nullptr)
{
    rt_asph_terms<NA> A; /* read once, before the iteration */
    rt_asph_read<NA>(S, flags, A);
    bool live[R];
    double res[R];
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = -y[r][2] / u[r][2];
        res[r] = RT_NAN;
        live[r] = true;
        any = true;
    }
    if constexpr (CENSUS) {
        if (RT_WAVE_ANY(any))
            ++census[2]; /* a solve this wavefront enters */
    }
#pragma unroll 1
    for (int itr = 0; itr < 5; ++itr) {
        if (!RT_WAVE_ANY(any))
            break;
        any = false;
        if constexpr (CENSUS)
            ++census[0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r])
                continue;
            if constexpr (CENSUS)
                ++census[1];
            if (rt_newton_iterate_t<NA>(A, flags, y[r], u[r], s[r], res[r]) ||
                s[r] != s[r])
                live[r] = false;
            else
                any = true;
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        s[r] = res[r];
}

/* ------------------------------------------------------------------ */
/* default arithmetic for even aspheres (RT_F_FAST)                   */
/* ------------------------------------------------------------------ */
/*
 * The contract for iterated aspheres is 1e-8 relative (BASELINE north_star),
 * not bit identity, and the exact Newton solve above is FP64-issue bound:
 * three IEEE divisions and a square root per iterate, each a 10-20
 * instruction sequence.  With RT_F_FAST (the default since round 3;
 * rt_set_option "exact_asphere" clears it) an
 * aspheric element runs the same iteration -- same start, same |step| <= 1e-7
 * test, same five-iterate limit, same NaN on failure
 * (rayopt/elements.py:333-349 + scipy newton) -- on
 *   * fused multiply-adds (Horner for sag and slope in one pass, coefficients
 *     zero-padded to a fixed term count so they sit in SGPRs for the whole
 *     solve instead of being re-read per iterate),
 *   * v_rcp_f64 / v_rsq_f64 (2^-23) plus Newton-Raphson refinement instead of
 *     IEEE division and sqrt,
 *   * ONE reciprocal per iterate: with root = sqrt(1 - (1+k) c^2 r^2),
 *     A = 1 + root,
 *         fval       = pz - c r^2/A - poly          (surface_sag, :440-455)
 *         fder       = (px ux + py uy) e + uz,  e = -c/root - dpoly (:457-475)
 *         fval A     = (pz - poly) A - c r^2                       =: num
 *         fder root  = uz root - (px ux + py uy)(c + dpoly root)   =: den
 *         fval/fder  = num root / (A den)
 *     (fval == 0 <=> num == 0 and fder == 0 <=> den == 0 since A >= 1 and
 *     root > 0; a point with root == 0 exactly -- the rim of the base conic --
 *     is handed to rt_newton_iterate: there the reference's result is decided
 *     by 0 * inf and inf - inf in its normal).
 * Results agree with the exact path to ~1e-13 (tests: <= 1e-9 asserted,
 * identical NaN masks on the asphere goldens).
 *
 * The same primitives are NOT offered for planes, spheres and conics: tried
 * (round 2), the closed form on FMA / rcp / rsq stays ~1e-14 from the exact
 * path on ordinary systems but reached 1.9e-9 on the worst-conditioned of
 * the random tilted systems (tests/golden/tilted_seed_4200) -- outside the
 * 1e-10 contract of the closed form, which the exact path meets there only
 * because it repeats numpy's operations in numpy's order.
 */
RT_HD double rt_fma(double a, double b, double c)
{
    return __builtin_fma(a, b, c);
}

/* |1 - (1+k) c^2 r^2| below this: the point is on the rim of the base conic
 * as far as rounding can tell -- there the exact arithmetic decides */
#define RT_RIM 1e-12

/* 1/x: hardware seed + NR steps (each squares the relative error) */
template <int STEPS>
RT_HD double rt_rcp_fast(double x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    double r = __builtin_amdgcn_rcp(x);
#else
    double r = 1. / x;
#endif
    double q = r;
#pragma unroll
    for (int k = 0; k < STEPS; ++k)
        q = rt_fma(q, rt_fma(-x, q, 1.), q);
    /* 1/0 = +-inf and 1/inf = 0 as IEEE division has them: the refinement
     * would turn the seed's inf / 0 into NaN */
    return (x == 0. || x - x != 0.) ? r : q;
}

/* sqrt(w) and 1/sqrt(w) together (coupled Goldschmidt steps on the hardware
 * rsq seed); w == 0 -> (0, inf), w < 0 or NaN -> NaN like sqrt */
template <int STEPS>
RT_HD void rt_sqrt_fast(double w, double &root, double &rinv)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const double r0 = __builtin_amdgcn_rsq(w);
#else
    const double r0 = 1. / sqrt(w);
#endif
    double g = w * r0, h = .5 * r0;
#pragma unroll
    for (int k = 0; k < STEPS; ++k) {
        const double e = rt_fma(-h, g, .5);
        g = rt_fma(g, e, g);
        h = rt_fma(h, e, h);
    }
    const bool zero = w == 0.;
    root = zero ? 0. : g;
    rinv = zero ? r0 : 2. * h;
}

/* sum_i a[i] r2^(i+1) and sum_i da[i] r2^i, NA terms each (zero padded) */
template <int NA>
RT_HD void rt_poly_fast(const double (&a)[NA], const double (&da)[NA],
                        double r2, double &poly, double &dpoly)
{
    double p = a[NA - 1], d = da[NA - 1];
#pragma unroll
    for (int i = NA - 2; i >= 0; --i) {
        p = rt_fma(p, r2, a[i]);
        d = rt_fma(d, r2, da[i]);
    }
    poly = p * r2;
    dpoly = d;
}

template <int R, int NA, bool CENSUS = false>
RT_HD void rt_newton_fast(const rt_surface *__restrict__ S, unsigned flags,
                          const double (&y)[R][3], const double (&u)[R][3],
                          double (&s)[R], unsigned *census = nullptr)
{
    /* wave-uniform: read once, live in SGPRs across the iteration */
    double a[NA], da[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a[i] = S->asph[i];
        da[i] = S->dasph[i];
    }
    const bool curved = flags & RT_F_CURVED;
    const double c = curved ? S->c : 0., kc2 = curved ? S->kc2 : 0.;
    bool live[R];
    double res[R];
    bool any = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = -y[r][2] * rt_rcp_fast<2>(u[r][2]);
        res[r] = RT_NAN;
        live[r] = true;
        any = true;
    }
    if constexpr (CENSUS) {
        if (RT_WAVE_ANY(any))
            ++census[2]; /* a solve this wavefront enters */
    }
#pragma unroll 1
    for (int itr = 0; itr < 5; ++itr) {
        if (!RT_WAVE_ANY(any))
            break;
        any = false;
        if constexpr (CENSUS)
            ++census[0];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (!live[r])
                continue;
            if constexpr (CENSUS)
                ++census[1];
            const double px = rt_fma(s[r], u[r][0], y[r][0]);
            const double py = rt_fma(s[r], u[r][1], y[r][1]);
            const double pz = rt_fma(s[r], u[r][2], y[r][2]);
            const double r2 = rt_fma(px, px, py * py);
            double root = 1., rinv;
            if (curved) {
                const double w = rt_fma(-kc2, r2, 1.);
                if (__builtin_fabs(w) < RT_RIM) {
                    /* on the rim of the base conic -- sqrt(1 - (1+k) c^2
                     * r^2) zero or a rounding error away from it -- the
                     * reference's result hangs on IEEE corners (normal =
                     * 0 * inf or inf - inf: NaN; else an infinite derivative
                     * and a zero step) and on whether ITS argument of the
                     * root rounds to zero: this iterate is evaluated its
                     * way, operation for operation
                     * (tests/golden/asphere_newton_rim) */
                    if (rt_newton_iterate(S, flags, y[r], u[r], s[r], res[r]))
                        live[r] = false;
                    else
                        any = true;
                    continue;
                }
                rt_sqrt_fast<1>(w, root, rinv);
            }
            double poly, dpoly;
            rt_poly_fast<NA>(a, da, r2, poly, dpoly);
            const double A = 1. + root;
            const double num = rt_fma(pz - poly, A, -(c * r2));
            if (num == 0.) { /* fval == 0: the current iterate is the root */
                res[r] = s[r];
                live[r] = false;
                continue;
            }
            const double dotxy = rt_fma(px, u[r][0], py * u[r][1]);
            const double den =
                rt_fma(u[r][2], root, -(dotxy * rt_fma(dpoly, root, c)));
            if (den == 0.) { /* "Derivative was zero" -> NaN */
                live[r] = false;
                continue;
            }
            const double step = (num * root) * rt_rcp_fast<1>(A * den);
            const double p = s[r] - step;
            if (rt_isclose(p, s[r], 1e-7)) {
                res[r] = p;
                live[r] = false;
                continue;
            }
            s[r] = p;
            if (p == p)
                any = true;
            else
                live[r] = false; /* NaN stays NaN: see rt_newton */
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        s[r] = res[r];
}

/* clip + refract/reflect at the intercept y: iv -> u (elements.py:309-314),
 * one ray, every operation as the reference's */
RT_HD void rt_bend_ray(const rt_surface *__restrict__ S, unsigned flags,
                       int clip, const double (&y)[3], const double (&iv)[3],
                       double (&u)[3])
{
    u[0] = iv[0];
    u[1] = iv[1];
    u[2] = iv[2];
    const double rxy = y[0] * y[0] + y[1] * y[1];
    if (clip) {
        /* Element.clip: NaN-poison the direction outside the aperture;
         * NaN coordinates compare false and are poisoned as well */
        if (!(rxy <= S->radius2))
            u[0] = u[1] = u[2] = RT_NAN;
    }
    if (flags & RT_F_REFRACT) {
        /* Spencer & Murty; q = (x e, y e, 1) un-normalised normal */
        double qx, qy;
        if (flags & (RT_F_CURVED | RT_F_ASPH)) {
            const double e = rt_normal_e(S, flags, rxy,
                                         rt_conic_root(S, flags, rxy));
            qx = y[0] * e;
            qy = y[1] * e;
        } else {
            qx = 0.;
            qy = 0.;
        }
        const double r2 = (qx * qx + qy * qy) + 1.;
        const double dot = (u[0] * qx + u[1] * qy) + u[2] * 1.;
        const double num = S->muf * dot;
        double a, g = 0.;
        /* r2 >= 1 or NaN: only its upper end needs the check */
        if ((flags & RT_F_RANGE) &&
            !RT_WAVE_ANY(r2 >= RT_RANGE_BIG || RT_ODD_MAG(num))) {
            const double rr = rt_rcp_refined(r2);
            a = rt_quot(num, r2, rr);
            if (!(flags & RT_F_MIRROR)) {
                /* in range a*a and b are both >= 2^-400 in magnitude:
                 * their difference is 0 or >= 2^-453, never inside
                 * (0, 2^-767) */
                const double b = rt_quot(S->mu2m1, r2, rr);
                g = -a + S->smu * rt_sqrt_unscaled(a * a - b);
            }
        } else {
            a = num / r2;
            if (!(flags & RT_F_MIRROR)) {
                const double b = S->mu2m1 / r2;
                g = -a + S->smu * sqrt(a * a - b);
            }
        }
        if (flags & RT_F_MIRROR) {
            const double a2 = 2. * a;
            u[0] = u[0] - a2 * qx;
            u[1] = u[1] - a2 * qy;
            u[2] = u[2] - a2 * 1.;
        } else {
            u[0] = S->muf * u[0] + g * qx;
            u[1] = S->muf * u[1] + g * qy;
            u[2] = S->muf * u[2] + g * 1.;
        }
    }
}

/* clip + refraction at an aspheric element on the fast arithmetic: the same
 * Spencer & Murty update as rt_step_bend (elements.py:351-369) */
template <int R, int NA>
RT_HD void rt_bend_fast(const rt_surface *__restrict__ S, unsigned flags,
                        int clip, const double (&y)[R][3],
                        const double (&iv)[R][3], double (&u)[R][3])
{
    double a[NA], da[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        a[i] = S->asph[i];
        da[i] = S->dasph[i];
    }
    const bool curved = flags & RT_F_CURVED;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        u[r][0] = iv[r][0];
        u[r][1] = iv[r][1];
        u[r][2] = iv[r][2];
        const double rxy = y[r][0] * y[r][0] + y[r][1] * y[r][1];
        if (clip) {
            if (!(rxy <= S->radius2))
                u[r][0] = u[r][1] = u[r][2] = RT_NAN;
        }
        if (flags & RT_F_REFRACT) {
            double root = 1., rinv = 1., poly, dpoly;
            if (curved) {
                const double w = rt_fma(-S->kc2, rxy, 1.);
                if (__builtin_fabs(w) < RT_RIM) {
                    /* on the rim of the base conic the normal is 0 * inf or
                     * inf in the reference: its operations decide */
                    rt_bend_ray(S, flags, clip, y[r], iv[r], u[r]);
                    continue;
                }
                rt_sqrt_fast<2>(w, root, rinv);
            }
            rt_poly_fast<NA>(a, da, rxy, poly, dpoly);
            const double e = -(curved ? rt_fma(S->c, rinv, dpoly) : dpoly);
            const double qx = y[r][0] * e, qy = y[r][1] * e;
            const double r2 = rt_fma(qx, qx, rt_fma(qy, qy, 1.));
            const double inv = rt_rcp_fast<2>(r2);
            const double dot = rt_fma(u[r][0], qx, rt_fma(u[r][1], qy, u[r][2]));
            const double am = S->muf * dot * inv;
            if (flags & RT_F_MIRROR) {
                const double a2 = 2. * am;
                u[r][0] = rt_fma(-a2, qx, u[r][0]);
                u[r][1] = rt_fma(-a2, qy, u[r][1]);
                u[r][2] = u[r][2] - a2;
            } else {
                const double b = S->mu2m1 * inv;
                double sq, sinv;
                rt_sqrt_fast<2>(rt_fma(am, am, -b), sq, sinv);
                const double g = rt_fma(S->smu, sq, -am);
                u[r][0] = rt_fma(S->muf, u[r][0], g * qx);
                u[r][1] = rt_fma(S->muf, u[r][1], g * qy);
                u[r][2] = rt_fma(S->muf, u[r][2], g);
            }
        }
    }
}

/* term count classes: the coefficient arrays are zero beyond nasph */
#define RT_FAST_DISPATCH(call4, call7, call10)                                \
    do {                                                                      \
        if (S->nasph <= 4) {                                                  \
            call4;                                                            \
        } else if (S->nasph <= 7) {                                           \
            call7;                                                            \
        } else {                                                              \
            call10;                                                           \
        }                                                                     \
    } while (0)

/*
 * Ray length to the element's surface: Spheroid.intercept (elements.py:
 * 477-501) -- Newton for aspheres, plane, or the closed-form conic root.
 */
template <int R, bool CENSUS = false>
RT_HD void rt_intercept(const rt_surface *__restrict__ S, unsigned flags,
                        const double (&y)[R][3], const double (&iv)[R][3],
                        double (&s)[R], unsigned *census = nullptr)
{
    if (flags & RT_F_FAST) {
        RT_FAST_DISPATCH(
            (rt_newton_fast<R, 4, CENSUS>(S, flags, y, iv, s, census)),
            (rt_newton_fast<R, 7, CENSUS>(S, flags, y, iv, s, census)),
            (rt_newton_fast<R, RT_MAX_ASPH, CENSUS>(S, flags, y, iv, s,
                                                    census)));
    } else if (flags & RT_F_ASPH) {
        RT_FAST_DISPATCH(
            (rt_newton<R, 4, CENSUS>(S, flags, y, iv, s, census)),
            (rt_newton<R, 7, CENSUS>(S, flags, y, iv, s, census)),
            (rt_newton<R, RT_MAX_ASPH, CENSUS>(S, flags, y, iv, s, census)));
    } else if (!(flags & RT_F_CURVED)) {
#pragma unroll
        for (int r = 0; r < R; ++r)
            s[r] = -y[r][2] / iv[r][2];
    } else {
        const double c = S->c;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            double uy, uu, yy;
            if (!(flags & RT_F_CONIC)) {
                uy = (iv[r][0] * y[r][0] + iv[r][1] * y[r][1]) +
                     iv[r][2] * y[r][2];
                uu = 1.;
                yy = (y[r][0] * y[r][0] + y[r][1] * y[r][1]) +
                     y[r][2] * y[r][2];
            } else {
                const double kw = S->kw;
                uy = ((iv[r][0] * y[r][0]) * 1. + (iv[r][1] * y[r][1]) * 1.) +
                     (iv[r][2] * y[r][2]) * kw;
                uu = ((iv[r][0] * iv[r][0]) * 1. +
                      (iv[r][1] * iv[r][1]) * 1.) +
                     (iv[r][2] * iv[r][2]) * kw;
                yy = ((y[r][0] * y[r][0]) * 1. + (y[r][1] * y[r][1]) * 1.) +
                     (y[r][2] * y[r][2]) * kw;
            }
            const double d = c * uy - iv[r][2];
            const double e = c * uu;
            const double f = c * yy - 2. * y[r][2];
            const double w = d * d - e * f;
            double g;
            if (RT_WAVE_ANY(w > 0. && w < RT_SQRT_SCALED_BELOW))
                g = sqrt(w);
            else
                g = rt_sqrt_unscaled(w);
            if (flags & RT_F_ALT)
                g *= -1.;
            const double num = -(d + g);
            /* e = c * 1. = c on a sphere: wave-uniform, its refined
             * reciprocal comes with the table */
            if ((flags & (RT_F_RANGE | RT_F_CONIC)) != RT_F_RANGE ||
                RT_WAVE_ANY(RT_ODD_MAG(num)))
                s[r] = num / e;
            else
                s[r] = rt_quot(num, c, S->rc);
        }
    }
}

/*
 * One launch ray of field F through pupil coordinates (px, py):
 * Infinite/FiniteConjugate.aim after the per-field frame has been built by
 * the host (conjugates.py:137-166, 236-255; Pupil.map pupils.py:97-107).
 */
template <bool ASPH = true>
RT_HD void rt_generate_ray(const rt_field *__restrict__ F, double px,
                           double py, const rt_surface *__restrict__ S0,
                           double (&y)[1][3], double (&u)[1][3])
{
    px *= F->am;
    py *= F->am;
    if (!F->finite) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u[0][c] = F->u[c];
            y[0][c] = F->base[c] + (px * F->s[c] + py * F->m[c]);
        }
        double t[1]; /* y += surface.intercept(y, u) u  (:253-254) */
        rt_intercept<1>(S0, ASPH ? S0->flags
                             : S0->flags & ~(RT_F_ASPH | RT_F_FAST), y, u, t);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            y[0][c] = y[0][c] + t[0] * u[0][c];
    } else {
        const double qx = F->z * tan(px), qy = F->z * tan(py);
        double d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            y[0][c] = F->base[c];
            d[c] = F->u[c] + (qx * F->s[c] + qy * F->m[c]);
        }
        const double nrm = sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            u[0][c] = d[c] / nrm;
            if (F->flip)
                u[0][c] *= -1.;
        }
    }
}

/*
 * One element for R rays.  In: y,u in the global orientation relative to the
 * previous vertex (what System.propagate carries between elements).  Out:
 * y = intercept, u = outgoing direction, iv = incoming direction, t = OPL,
 * all in the element-normal frame (the tuple System.propagate yields).
 */
template <int R, bool CENSUS = false>
RT_HD void rt_step_hit(const rt_surface *__restrict__ S, unsigned flags,
                       double (&y)[R][3], const double (&u)[R][3],
                       double (&iv)[R][3], double (&t)[R],
                       unsigned *census = nullptr)
{
    /* transfer: y - e.offset, then to_normal (system.py:461) */
#pragma unroll
    for (int r = 0; r < R; ++r) {
        y[r][0] -= S->offset[0];
        y[r][1] -= S->offset[1];
        y[r][2] -= S->offset[2];
        iv[r][0] = u[r][0];
        iv[r][1] = u[r][1];
        iv[r][2] = u[r][2];
        if (flags & RT_F_ROTATED) {
            rt_rot_to(S->rot, y[r]);
            rt_rot_to(S->rot, iv[r]);
        }
    }

    /* intercept */
    double s[R];
    rt_intercept<R, CENSUS>(S, flags, y, iv, s, census);

#pragma unroll
    for (int r = 0; r < R; ++r) {
        /* y = y0 + t*u0 */
        y[r][0] = y[r][0] + s[r] * iv[r][0];
        y[r][1] = y[r][1] + s[r] * iv[r][1];
        y[r][2] = y[r][2] + s[r] * iv[r][2];
        t[r] = s[r] * S->n0;
    }
}

/* clip + refract/reflect at the intercept y: iv -> u for R rays */
template <int R>
RT_HD void rt_step_bend(const rt_surface *__restrict__ S, unsigned flags,
                        int clip, const double (&y)[R][3],
                        const double (&iv)[R][3], double (&u)[R][3])
{
    if (flags & RT_F_FAST) {
        RT_FAST_DISPATCH((rt_bend_fast<R, 4>(S, flags, clip, y, iv, u)),
                         (rt_bend_fast<R, 7>(S, flags, clip, y, iv, u)),
                         (rt_bend_fast<R, RT_MAX_ASPH>(S, flags, clip, y, iv,
                                                       u)));
        return;
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
        rt_bend_ray(S, flags, clip, y[r], iv[r], u[r]);
}

template <int R>
RT_HD void rt_step(const rt_surface *__restrict__ S, unsigned flags, int clip,
                   double (&y)[R][3], double (&u)[R][3], double (&iv)[R][3],
                   double (&t)[R])
{
    rt_step_hit<R>(S, flags, y, u, iv, t);
    rt_step_bend<R>(S, flags, clip, y, iv, u);
}

/* from_normal of the element just left (system.py:464); y,u copies */
template <int R>
RT_HD void rt_leave(const rt_surface *__restrict__ S, unsigned flags,
                    double (&y)[R][3], double (&u)[R][3])
{
    if (flags & RT_F_ROTATED) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rt_rot_from(S->rot, y[r]);
            rt_rot_from(S->rot, u[r]);
        }
    }
}

#endif /* RT_MATH_H */
