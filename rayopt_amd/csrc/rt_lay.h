/*
 * rt_lay.h -- where the result arrays live, as the kernels see it.
 *
 * The documented layout (include/rt_mi355.h), "SoA":
 *     Y,U,I [L][3][ld], T [L][ld]      cs = ld, ss = 3 ld, ssT = ld
 * and a ray's column is its index j.
 *
 * Laboratory build only (-DRT_BUILD_PROBES, librt_mi355_probes.so,
 * rt_set_option "tile_rays" = TR): the batch is cut into tiles of TR rays and
 * a tile holds ALL its rows back to back, [tile][L][10][TR] with the ten
 * components y0 y1 y2 u0 u1 u2 i0 i1 i2 t, so that a workgroup's whole output
 * is one contiguous region: cs = TR, ss = ssT = 10 TR, tile stride
 * ts = L 10 TR; element (array, s, c) of ray j is at
 *     base[array] + s*ss + c*cs + (j >> tshift)*ts + (j & (TR - 1)).
 * Measured no better than SoA (profiles/HISTORY.md); the shipped library
 * addresses SoA only.
 */
#ifndef RT_LAY_H
#define RT_LAY_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct rt_lay {
    double *Y, *U, *I, *T;
    int64_t cs, ss, ssT;
#ifdef RT_BUILD_PROBES
    int64_t ts;
    int tshift;
#endif
};

__device__ __forceinline__ int64_t rt_col(const rt_lay &a, int64_t j)
{
#ifdef RT_BUILD_PROBES
    return (j >> a.tshift) * a.ts + (j & (((int64_t)1 << a.tshift) - 1));
#else
    (void)a;
    return j;
#endif
}

#endif /* RT_LAY_H */
