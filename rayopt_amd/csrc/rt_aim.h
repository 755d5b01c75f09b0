/*
 * rt_aim.h -- aiming of one field point, start to finish, as host+device
 * code: launch frame, one-ray trace without stores, secant / regula falsi.
 * The kernel (rt_kernels.h) runs it one lane per field; tests/hostemu runs
 * the same functions on the CPU.  Reference: System.pupil, _aim_pupil,
 * aim_chief, aim_marginal (rayopt/system.py:507-593); frames: Conjugate.aim
 * (rayopt/conjugates.py:137-166, 236-255), sagittal_meridional
 * (rayopt/utils.py:106-114).
 */
#ifndef RT_AIM_H
#define RT_AIM_H

#include "rt_math.h"

RT_HD void rt_cross(const double (&a)[3], const double (&b)[3],
                    double (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

RT_HD void rt_unit(double (&v)[3])
{
    const double n = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    v[0] /= n;
    v[1] /= n;
    v[2] /= n;
}

/*
 * Launch frame of one field for pupil distance z and (scalar) aperture a:
 * what rayopt_amd/launch.py: field_frames hands rt_generate_rays.
 */
RT_HD void rt_field_frame(const rt_aim_seed *sd, double z, double a,
                          rt_field *F)
{
    double u[3];
    F->z = z;
    F->finite = sd->finite;
    F->flip = 0;
    if (!sd->finite) {
        for (int c = 0; c < 3; ++c)
            u[c] = sd->dir[c];
        F->am = fabs(a);
        F->base[0] = 0. - z * u[0];
        F->base[1] = 0. - z * u[1];
        F->base[2] = z - z * u[2];
    } else {
        u[0] = sd->telecentric ? 0. : 0. - sd->point[0];
        u[1] = sd->telecentric ? 0. : 0. - sd->point[1];
        u[2] = sd->telecentric ? z : z - sd->point[2];
        F->flip = z < 0;
        F->am = fabs(atan2(a, z));
        for (int c = 0; c < 3; ++c)
            F->base[c] = sd->point[c];
    }
    /* sagittal / meridional unit vectors about the axis (0, 0, z) */
    const double axis[3] = {0., 0., z};
    double s[3], m[3];
    rt_cross(u, axis, s);
    if (s[0] == 0 && s[1] == 0 && s[2] == 0) {
        s[0] = 1.;
        s[1] = s[2] = 0.;
    }
    rt_cross(u, s, m);
    rt_unit(s);
    rt_unit(m);
    for (int c = 0; c < 3; ++c) {
        F->u[c] = u[c];
        F->s[c] = s[c];
        F->m[c] = m[c];
    }
}

/*
 * One ray of field F through pupil coordinate (px, py), elements 1..last,
 * nothing stored.  Returns the intercept on element `last` (its frame) and,
 * over elements 1..last, the largest |y_xy|^2 / radius^2 - 1 (NaN if the ray
 * is lost on the way, like numpy's max).
 */
RT_HD void rt_aim_trace(const rt_surface *__restrict__ tab, int last,
                        const rt_field *F, double px, double py,
                        double (&hit)[2], double &fill_last,
                        double &fill_worst)
{
    double y[1][3], u[1][3], iv[1][3], t[1];
    rt_generate_ray(F, px, py, tab, y, u);
    rt_leave<1>(tab, tab->flags, y, u);
    double worst = -HUGE_VAL;
    bool lost = false;
    for (int s = 1; s <= last; ++s) {
        const rt_surface *S = tab + s;
        const unsigned flags = S->flags;
        rt_step<1>(S, flags, 0, y, u, iv, t);
        const double v =
            (y[0][0] * y[0][0] + y[0][1] * y[0][1]) / S->radius2 - 1;
        if (v != v)
            lost = true;
        else if (v > worst)
            worst = v;
        if (s == last) {
            hit[0] = y[0][0];
            hit[1] = y[0][1];
            fill_last = v;
        }
        rt_leave<1>(S, flags, y, u);
    }
    fill_worst = lost ? NAN : worst;
}

/* aim_chief: distance z0 + alpha a0 that puts the chief ray on the stop
 * centre; secant in alpha on (yo . y_stop) / stop radius */
RT_HD int rt_aim_chief(const rt_surface *__restrict__ tab,
                       const rt_aim_seed *sd, const rt_aim_args *g, double p,
                       double *z)
{
    const double z0 = sd->z0;
    *z = z0;
    /* np.isclose(yo, 0): on axis there is nothing to aim; nor for a
     * telecentric object, whose chief rays do not depend on the pupil
     * distance (aim_chief, system.py:509-510) */
    if ((fabs(sd->yo[0]) <= 1e-8 && fabs(sd->yo[1]) <= 1e-8) ||
        (sd->finite && sd->telecentric) || g->no_chief)
        return 0;
    const double rad = sqrt(tab[g->stop].radius2);
    rt_field F;
    double hit[2], fl, fw;
    double a0 = 0., a1 = 1e-4, f0, f1;
    rt_field_frame(sd, z0 + a0 * p, p, &F);
    rt_aim_trace(tab, g->stop, &F, 0., 0., hit, fl, fw);
    f0 = (sd->yo[0] * hit[0] + sd->yo[1] * hit[1]) / rad;
    rt_field_frame(sd, z0 + a1 * p, p, &F);
    rt_aim_trace(tab, g->stop, &F, 0., 0., hit, fl, fw);
    f1 = (sd->yo[0] * hit[0] + sd->yo[1] * hit[1]) / rad;
    for (int it = 0; it < g->maxiter; ++it) {
        const double step = f1 != f0 ? f1 * (a1 - a0) / (f1 - f0) : 0.;
        a0 = a1;
        f0 = f1;
        a1 = a1 - step;
        if (fabs(step) <= g->tol) {
            *z = z0 + a1 * p;
            return 0;
        }
        rt_field_frame(sd, z0 + a1 * p, p, &F);
        rt_aim_trace(tab, g->stop, &F, 0., 0., hit, fl, fw);
        f1 = (sd->yo[0] * hit[0] + sd->yo[1] * hit[1]) / rad;
    }
    *z = NAN;
    return 1;
}

/* aim_marginal: scale x of the aperture a0 at which the ray through pupil
 * coordinate (px, py) grazes the stop (or, rim, the first limiting aperture):
 * bracket by expansion, then regula falsi with the Illinois modification */
RT_HD int rt_aim_marginal(const rt_surface *__restrict__ tab, int nsurf,
                          const rt_aim_seed *sd, const rt_aim_args *g,
                          double z, double px, double py, double *x_out)
{
    const int last = g->rim ? nsurf - 2 : g->stop;
    const double p = sd->a0;
    rt_field F;
    double hit[2], fl, fw;
#define RT_MARGIN(scale, out)                                                 \
    do {                                                                      \
        rt_field_frame(sd, z, fabs((scale) * p), &F);                         \
        rt_aim_trace(tab, last, &F, px, py, hit, fl, fw);                     \
        (out) = g->rim ? fw : fl;                                             \
    } while (0)
    double lo = 0., flo, hi = 1., fhi = 0.;
    RT_MARGIN(1e-9, flo); /* ~ the chief ray: inside, < 0 */
    int it = 0;
    for (; it < g->maxiter; ++it) { /* expand until outside */
        RT_MARGIN(hi, fhi);
        const bool bad = fhi != fhi;
        const bool inside = !bad && fhi < 0;
        if (!bad && !inside)
            break;
        if (inside) {
            lo = hi;
            flo = fhi;
        }
        hi = bad ? hi / 2 : hi * (1 - fhi);
    }
    *x_out = NAN;
    if (it == g->maxiter)
        return 2;
    double x = 0., fx, side = 0.;
    for (it = 0; it < g->maxiter; ++it) {
        x = (lo * fhi - hi * flo) / (fhi - flo);
        if (!isfinite(x))
            x = (lo + hi) / 2;
        RT_MARGIN(x, fx);
        const bool neg = fx < 0;
        /* Illinois: halve the retained end's value when it is kept twice */
        if (neg) {
            if (side == -1.)
                fhi = fhi / 2;
            flo = fx;
            lo = x;
            side = -1.;
        } else {
            if (side == 1.)
                flo = flo / 2;
            fhi = fx;
            hi = x;
            side = 1.;
        }
        if (fabs(fx) <= g->tol) {
            *x_out = x * p;
            return 0;
        }
    }
#undef RT_MARGIN
    return 3;
}

/* System.pupil for one field: z, then a = [[-sag, -mer], [+sag, +mer]] */
RT_HD int rt_aim_field(const rt_surface *__restrict__ tab, int nsurf,
                       const rt_aim_seed *sd, const rt_aim_args *g, double *z,
                       double (&a)[2][2])
{
    for (int i = 0; i < 2; ++i)
        a[i][0] = a[i][1] = NAN;
    int rc = rt_aim_chief(tab, sd, g, fabs(sd->a0), z);
    if (rc)
        return rc;
    for (int axis = 1; axis >= 0; --axis)
        for (int sign = 1; sign >= 0; --sign) {
            const double e = 2 * sign - 1.;
            double x;
            rc = rt_aim_marginal(tab, nsurf, sd, g, *z, axis == 0 ? e : 0.,
                                 axis == 1 ? e : 0., &x);
            if (rc)
                return rc;
            a[sign][axis] = e * fabs(x);
        }
    return 0;
}

#endif /* RT_AIM_H */
