/*
 * Plain-C host of the engine: traces a bundle through the biconvex singlet
 * of BASELINE config C1 using nothing but include/rt_mi355.h -- what a
 * non-Python binding (cgo, JNI, N-API ...) would do.  Built and run by
 * tests/test_cabi_gpu.py::test_c_example.
 *
 *   hipcc/gcc -Iinclude examples/trace_singlet.c -Lrayopt_amd -lrt_mi355 -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "rt_mi355.h"

static void element(rt_surface *s, double roc, double distance, double n0,
                    double n, double radius)
{
    memset(s, 0, sizeof *s);
    const double c = roc != 0. ? 1. / roc : 0.;
    const double mu = n > 0. ? n0 / n : 1.; /* n <= 0: no material */
    s->c = c;
    s->k = 0.;
    s->kw = 1.;
    s->kc2 = c * c;
    s->radius2 = radius * radius;
    s->mu = mu;
    s->muf = fabs(mu);
    s->smu = mu > 0. ? 1. : -1.;
    s->mu2m1 = mu * mu - 1.;
    s->n0 = n0;
    s->offset[2] = distance;
    s->rot[0] = s->rot[4] = s->rot[8] = 1.;
    s->flags = (c != 0. ? RT_F_CURVED : 0) | (mu != 1. ? RT_F_REFRACT : 0);
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? atoll(argv[1]) : 100000;
    rt_surface sys[4];
    element(&sys[0], 0., 0., 1., 1., INFINITY);       /* object, air */
    element(&sys[1], 51.5, 10., 1., 1.5168, 10.);     /* front */
    element(&sys[2], -51.5, 5., 1.5168, 1., 10.);     /* back */
    element(&sys[3], 0., 48.2, 1., -1., 8.);          /* image, no material */

    rt_ctx *ctx = NULL;
    if (rt_create(0, &ctx) != RT_OK) {
        fprintf(stderr, "rt_create: %s\n", rt_last_error(NULL));
        return 2;
    }
    double *y = calloc(3 * n, sizeof(double)), *u = calloc(3 * n, sizeof(double));
    for (int64_t j = 0; j < n; ++j) { /* a fan along y, collimated */
        y[3 * j + 1] = -9.5 + 19. * (double)j / (double)(n - 1);
        u[3 * j + 2] = 1.;
    }
    int rc = rt_upload_system(ctx, sys, 4);
    if (rc == RT_OK) rc = rt_set_rays(ctx, y, u, n, RT_LAYOUT_AOS);
    if (rc == RT_OK) rc = rt_trace(ctx, 1, 0, 1);
    double ms = 0., rms = 0.;
    if (rc == RT_OK) rc = rt_kernel_ms(ctx, &ms);
    double *img = malloc(3 * n * sizeof(double));
    if (rc == RT_OK) rc = rt_download(ctx, RT_Y, 3, 4, img); /* (3, n) SoA */
    if (rc == RT_OK) rc = rt_rms(ctx, 3, -1, &rms);
    if (rc != RT_OK) {
        fprintf(stderr, "error %d: %s\n", rc, rt_last_error(ctx));
        return 1;
    }
    /* paraxial check: a ray at height h crosses the axis near the focus */
    const int64_t mid = n / 2 + n / 20;
    printf("rays %lld kernel_ms %.4f rms %.6f y_image[mid] %.17g\n",
           (long long)n, ms, rms, img[n + mid]);
    rt_destroy(ctx);
    free(y); free(u); free(img);
    return 0;
}
