/*
 * rt_lay.h -- where the result arrays live, as the kernels see it.
 *
 * The documented layout (include/rt_mi355.h), "SoA":
 *     Y,U,I [L][3][ld], T [L][ld]      cs = ld, ss = 3 ld, ssT = ld
 * and a ray's column is its index j.
 *
 * A batch whose rows would lie too far apart is cut into BLOCKS of bs rays
 * (rt_reserve: above 8.1*10^6 rays for 13 elements), every block with its
 * own Y | U | I | T planes in that same layout, bs the distance between
 * rows, ts doubles from one block to the next:
 *     element (array, s, c) of ray j  =
 *         base[array] + (j / bs) ts + s ss + c cs + j % bs,   cs = bs.
 * One block (ts = 0) is the documented layout itself.  Why: the 7-10 rows an
 * element writes at once, times the elements, are concurrent streams `cs`
 * doubles apart; when they reach over more than ~8-10 one-GiB regions of the
 * address space the device's address translation falls behind (10^8 rays:
 * 0.72 of the HBM spec as one block, 0.84 in blocks).
 *
 * Windows (rt_trace_chunk): j0 is added to the ray index, the arrays are not
 * shifted.
 *
 * Laboratory build only (-DRT_BUILD_PROBES, rt_set_option "tile_rays" = TR):
 * the same fields describe tile-major layouts, e.g. [tile][L][10][TR] with
 * the ten components y0 y1 y2 u0 u1 u2 i0 i1 i2 t (cs = TR, ss = ssT = 10 TR,
 * bs = TR, ts = L 10 TR).  Measured no better than SoA (profiles/HISTORY.md).
 */
#ifndef RT_LAY_H
#define RT_LAY_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct rt_lay {
    double *Y, *U, *I, *T;
    int64_t cs, ss, ssT;
    int64_t bs, ts; /* rays per block, doubles between blocks (0: one block) */
    int64_t j0;     /* first ray of the window the launch covers */
};

/* where ray j of a row lies relative to the row's start in block 0 */
__host__ __device__ __forceinline__ int64_t rt_block_col(int64_t bs, int64_t ts,
                                                         int64_t j)
{
    if (!ts) /* wave-uniform: one block, the documented layout */
        return j;
    const uint64_t b = (uint64_t)j / (uint64_t)bs;
    return (int64_t)b * ts + (j - (int64_t)b * bs);
}

__device__ __forceinline__ int64_t rt_col(const rt_lay &a, int64_t j)
{
    return rt_block_col(a.bs, a.ts, j + a.j0);
}

#endif /* RT_LAY_H */
