/*
 * TEST INFRASTRUCTURE ONLY.  Compiles the kernel's per-ray arithmetic
 * (rayopt_amd/csrc/rt_math.h, the very header the gfx950 kernel inlines) for
 * the host with g++, so the arithmetic can be checked against the golden
 * vectors in a container that has no GPU.  The product never loads this.
 */
#include <stdint.h>
#include "../../rayopt_amd/csrc/rt_math.h"
#include "../../rayopt_amd/csrc/rt_aim.h"

template <int R>
static void run(const rt_surface *surf, int start, int stop, int clip,
                const double *y0, const double *u0, int64_t n, double *Y,
                double *U, double *I, double *T)
{
    for (int64_t j0 = 0; j0 < n; j0 += R) {
        double y[R][3], u[R][3], iv[R][3], t[R];
        for (int r = 0; r < R; ++r) {
            const int64_t j = j0 + r < n ? j0 + r : n - 1; /* pad: repeat */
            for (int c = 0; c < 3; ++c) {
                y[r][c] = y0[j * 3 + c];
                u[r][c] = u0[j * 3 + c];
            }
        }
        rt_leave<R>(surf + (start - 1), surf[start - 1].flags, y, u);
        for (int s = start; s < stop; ++s) {
            const rt_surface *S = surf + s;
            rt_step<R>(S, S->flags, clip, y, u, iv, t);
            for (int r = 0; r < R; ++r) {
                const int64_t j = j0 + r;
                if (j >= n)
                    continue;
                const int64_t row = (int64_t)(s - start) * n + j;
                for (int c = 0; c < 3; ++c) {
                    Y[row * 3 + c] = y[r][c];
                    U[row * 3 + c] = u[r][c];
                    I[row * 3 + c] = iv[r][c];
                }
                T[row] = t[r];
            }
            rt_leave<R>(S, S->flags, y, u);
        }
    }
}

extern "C" int emu_trace(const rt_surface *surf, int start, int stop, int clip,
                         int rays_per_lane, const double *y0, const double *u0,
                         int64_t n, double *Y, double *U, double *I, double *T)
{
    switch (rays_per_lane) {
    case 1: run<1>(surf, start, stop, clip, y0, u0, n, Y, U, I, T); break;
    case 2: run<2>(surf, start, stop, clip, y0, u0, n, Y, U, I, T); break;
    case 4: run<4>(surf, start, stop, clip, y0, u0, n, Y, U, I, T); break;
    default: return -1;
    }
    return 0;
}

extern "C" int emu_generate(const rt_field *fields, int nfields,
                            const double *pupil, int64_t npupil,
                            const rt_surface *s0, double *Y, double *U)
{
    for (int f = 0; f < nfields; ++f)
        for (int64_t p = 0; p < npupil; ++p) {
            double y[1][3], u[1][3];
            rt_generate_ray(fields + f, pupil[2 * p], pupil[2 * p + 1], s0, y,
                            u);
            const int64_t r = (int64_t)f * npupil + p;
            for (int c = 0; c < 3; ++c) {
                Y[r * 3 + c] = y[0][c];
                U[r * 3 + c] = u[0][c];
            }
        }
    return 0;
}

extern "C" int emu_field_frame(const rt_aim_seed *seed, double z, double a,
                               rt_field *out)
{
    rt_field_frame(seed, z, a, out);
    return 0;
}

extern "C" int emu_aim_pupil(const rt_surface *tab, int nsurf,
                             const rt_aim_seed *seeds, int nfields,
                             const rt_aim_args *args, double *z, double *a,
                             int32_t *status)
{
    for (int f = 0; f < nfields; ++f) {
        double af[2][2];
        status[f] = rt_aim_field(tab + (int64_t)seeds[f].group * nsurf, nsurf,
                                 seeds + f, args, z + f, af);
        for (int i = 0; i < 2; ++i)
            for (int k = 0; k < 2; ++k)
                a[(f * 2 + i) * 2 + k] = af[i][k];
    }
    return 0;
}
