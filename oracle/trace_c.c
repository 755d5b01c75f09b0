/*
 * TEST INFRASTRUCTURE ONLY -- plain-C restatement of the reference algorithm
 * for the hot path GeometricTrace.propagate (quartiq/rayopt), ray by ray.
 *
 * Independent of the HIP sources (it shares no code with
 * rayopt_amd/csrc/): written from the reference's methods, each step citing
 * the reference line it follows.  It exists next to the numpy oracle
 * (oracle/trace_numpy.py) as a second, differently-shaped implementation
 * (scalar, ray by ray, no array temporaries) and because it spreads over the
 * host cores with OpenMP (measured: 8e6 ray-surface-ops/s on one core, 1e8 on
 * 8), which lets the GPU tests compare EVERY ray of a full-size (10^7 rays)
 * trace instead of a subsample.  Pinned like the numpy oracle: bit-identical to the
 * reference's golden vectors -- plane/sphere/conic surfaces, tilted elements
 * and the restated scipy Newton loop of the aspheres (tests/test_oracle_c.py).
 * Compile with -ffp-contract=off (numpy never fuses a*b+c); the two places
 * where the reference goes through BLAS, which does, spell their fma() out.
 * Never linked into or called by the product.
 *
 * Table layout: struct rt_surface of include/rt_mi355.h.
 */
#include <math.h>
#include <stdint.h>
#include "../include/rt_mi355.h"

/* Spheroid.surface_sag residual, rayopt/elements.py:440-455 */
static double sag(const rt_surface *s, const double p[3])
{
    double e = p[2];
    if (!(s->flags & (RT_F_CURVED | RT_F_ASPH)))
        return e;
    const double r2 = p[0] * p[0] + p[1] * p[1];
    if (s->flags & RT_F_CURVED)
        e -= s->c * r2 / (1 + sqrt(1 - s->kc2 * r2));
    if (s->flags & RT_F_ASPH) {
        double d = 0.;
        for (int i = s->nasph - 1; i >= 0; --i) {
            d += s->asph[i];
            d *= r2;
        }
        e -= d;
    }
    return e;
}

/* Spheroid.surface_normal, rayopt/elements.py:457-475: q = (x e, y e, 1) */
static void normal(const rt_surface *s, const double p[3], double q[3])
{
    q[0] = q[1] = 0.;
    q[2] = 1.;
    if (!(s->flags & (RT_F_CURVED | RT_F_ASPH)))
        return;
    const double r2 = p[0] * p[0] + p[1] * p[1];
    double e = 0.;
    if (s->flags & RT_F_CURVED)
        e -= s->c / sqrt(1 - s->kc2 * r2);
    if (s->flags & RT_F_ASPH) {
        double d = 0.;
        for (int i = s->nasph - 1; i >= 0; --i) {
            d *= r2;
            d += s->dasph[i];
        }
        e -= d;
    }
    q[0] = p[0] * e;
    q[1] = p[1] * e;
}

/* np.isclose(a, b, rtol=0, atol=tol) */
static int isclose(double a, double b, double tol)
{
    if (isfinite(a) && isfinite(b))
        return fabs(a - b) <= tol;
    return a == b;
}

/* Interface.intercept, rayopt/elements.py:333-349 + scipy newton (scalar
 * Newton-Raphson branch, tol=1e-7 absolute, maxiter=5, RuntimeError->NaN) */
static double newton_intercept(const rt_surface *s, const double y[3],
                               const double u[3])
{
    double p0 = -y[2] / u[2];
    for (int itr = 0; itr < 5; ++itr) {
        double x[3], q[3];
        for (int c = 0; c < 3; ++c)
            x[c] = y[c] + p0 * u[c];
        const double fval = sag(s, x);
        if (fval == 0)
            return p0;
        normal(s, x, q);
        /* np.dot of a (1,3) with a (3,1) array (:342): the BLAS chain */
        const double fder = fma(q[2], u[2], fma(q[1], u[1], q[0] * u[0]));
        if (fder == 0)
            return NAN;
        const double p = p0 - fval / fder;
        if (isclose(p, p0, 1e-7))
            return p;
        p0 = p;
    }
    return NAN;
}

/* Spheroid.intercept, rayopt/elements.py:477-501 */
static double intercept(const rt_surface *s, const double y[3],
                        const double u[3])
{
    if (s->flags & RT_F_ASPH)
        return newton_intercept(s, y, u);
    if (!(s->flags & RT_F_CURVED))
        return -y[2] / u[2];
    double uy, uu, yy;
    if (!(s->flags & RT_F_CONIC)) {
        uy = (u[0] * y[0] + u[1] * y[1]) + u[2] * y[2];
        uu = 1.;
        yy = (y[0] * y[0] + y[1] * y[1]) + y[2] * y[2];
    } else {
        const double k[3] = {1., 1., s->kw};
        uy = (u[0] * y[0] * k[0] + u[1] * y[1] * k[1]) + u[2] * y[2] * k[2];
        uu = (u[0] * u[0] * k[0] + u[1] * u[1] * k[1]) + u[2] * u[2] * k[2];
        yy = (y[0] * y[0] * k[0] + y[1] * y[1] * k[1]) + y[2] * y[2] * k[2];
    }
    const double d = s->c * uy - u[2];
    const double e = s->c * uu;
    const double f = s->c * yy - 2 * y[2];
    double g = sqrt(d * d - e * f);
    if (s->flags & RT_F_ALT)
        g *= -1;
    return -(d + g) / e;
}

/* v @ R (inverse: v @ R.T), rayopt/elements.py:156-175.  np.dot of an
 * (N,3) array with a 3x3 matrix is a BLAS dgemm: the inner index is summed
 * in order with fused multiply-adds (verified entry by entry against exact
 * rational arithmetic for numpy's OpenBLAS, N >= 2), hence the explicit
 * fma() here while everything else is compiled with -ffp-contract=off. */
static void rotate(const double r[9], int inverse, double v[3])
{
    double o[3];
    for (int j = 0; j < 3; ++j)
        o[j] = inverse ? fma(v[2], r[3 * j + 2],
                             fma(v[1], r[3 * j + 1], v[0] * r[3 * j]))
                       : fma(v[2], r[6 + j],
                             fma(v[1], r[3 + j], v[0] * r[j]));
    v[0] = o[0];
    v[1] = o[1];
    v[2] = o[2];
}

/*
 * GeometricTrace.propagate + System.propagate (geometric_trace.py:72-80,
 * system.py:459-464) for n rays given as (n,3) arrays in the normal frame of
 * element start-1.  Outputs (stop-start, n, 3) / (stop-start, n).
 */
int oracle_propagate(const rt_surface *tab, int start, int stop, int clip,
                     const double *y0, const double *u0, int64_t n, double *Y,
                     double *U, double *I, double *T)
{
    /* rays are independent: all host cores when built with -fopenmp */
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        double y[3], u[3];
        for (int c = 0; c < 3; ++c) {
            y[c] = y0[3 * r + c];
            u[c] = u0[3 * r + c];
        }
        if (tab[start - 1].flags & RT_F_ROTATED) { /* from_normal(init) */
            rotate(tab[start - 1].rot, 0, y);
            rotate(tab[start - 1].rot, 0, u);
        }
        for (int j = start; j < stop; ++j) {
            const rt_surface *s = tab + j;
            double i[3], q[3];
            for (int c = 0; c < 3; ++c) { /* y - e.offset; to_normal */
                y[c] -= s->offset[c];
                i[c] = u[c];
            }
            if (s->flags & RT_F_ROTATED) {
                rotate(s->rot, 1, y);
                rotate(s->rot, 1, i);
            }
            /* Interface.propagate, elements.py:306-315 */
            const double t = intercept(s, y, i);
            for (int c = 0; c < 3; ++c) {
                y[c] = y[c] + t * i[c];
                u[c] = i[c];
            }
            if (clip && !(y[0] * y[0] + y[1] * y[1] <= s->radius2))
                u[0] = u[1] = u[2] = NAN; /* Element.clip, :206-209 */
            if (s->flags & RT_F_REFRACT) { /* Interface.refract, :351-369 */
                normal(s, y, q);
                const double r2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
                const double a =
                    s->muf * ((u[0] * q[0] + u[1] * q[1]) + u[2] * q[2]) / r2;
                if (s->flags & RT_F_MIRROR) {
                    for (int c = 0; c < 3; ++c)
                        u[c] = u[c] - 2 * a * q[c];
                } else {
                    const double b = s->mu2m1 / r2;
                    const double g = -a + s->smu * sqrt(a * a - b);
                    for (int c = 0; c < 3; ++c)
                        u[c] = s->muf * u[c] + g * q[c];
                }
            }
            const int64_t row = (int64_t)(j - start) * n + r;
            for (int c = 0; c < 3; ++c) {
                Y[3 * row + c] = y[c];
                U[3 * row + c] = u[c];
                I[3 * row + c] = i[c];
            }
            T[row] = t * s->n0;
            if (s->flags & RT_F_ROTATED) { /* from_normal, system.py:464 */
                rotate(s->rot, 0, y);
                rotate(s->rot, 0, u);
            }
        }
    }
    return 0;
}
