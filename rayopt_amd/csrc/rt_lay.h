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
    /* the block of a whole 256-ray workgroup without a division (rt_col_wg):
     * blocks are whole workgroups and windows begin on workgroup borders, so
     * the block index is one number per workgroup, w / wgs with w = the
     * workgroup's index in the batch, formed on the scalar unit as
     * (w * wmagic) >> wshift.  wgs = 0: not available (rt_col divides) */
    uint32_t wgs, wmagic, wshift, w0;
};

#define RT_LAY_WG 256 /* rays per workgroup of the kernels that use rt_col_wg */

/*
 * Division of w < 2^24 by d <= 2^24 as a multiplication (Granlund &
 * Montgomery 1994, theorem 4.2 with N = 24): l = ceil(log2 d),
 * m = ceil(2^(24+l) / d) < 2^25, floor(w / d) = (w m) >> (24 + l).
 * 2^32 rays (more than 288 GB hold) are 2^24 workgroups.
 */
__host__ __device__ inline void rt_lay_set_window(rt_lay &a, int64_t lo,
                                                  int64_t ld)
{
    a.j0 = lo;
    a.wgs = a.wmagic = a.wshift = a.w0 = 0;
    if (!a.ts || a.bs % RT_LAY_WG || lo % RT_LAY_WG)
        return;
    const uint64_t d = (uint64_t)(a.bs / RT_LAY_WG);
    const uint64_t wmax = (uint64_t)((ld + RT_LAY_WG - 1) / RT_LAY_WG);
    if (d < 1 || d > ((uint64_t)1 << 24) || wmax >= ((uint64_t)1 << 24))
        return;
    uint32_t l = 0;
    while (((uint64_t)1 << l) < d)
        ++l;
    a.wshift = 24 + l;
    a.wmagic = (uint32_t)((((uint64_t)1 << a.wshift) + d - 1) / d);
    a.wgs = (uint32_t)d;
    a.w0 = (uint32_t)(lo / RT_LAY_WG);
}

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

/* the block of workgroup `wg` of the launch (its index in the batch: + w0) */
__host__ __device__ __forceinline__ uint64_t rt_wg_block(const rt_lay &a,
                                                         uint32_t wg)
{
    return ((uint64_t)(wg + a.w0) * a.wmagic) >> a.wshift;
}

/* rt_col for kernels whose workgroups are RT_LAY_WG consecutive rays, wg =
 * blockIdx.x: the block index comes from the scalar unit (a 64-bit division
 * per thread on the vector unit was 3 % of the headline trace's VALU
 * instructions), the thread adds one wave-uniform offset */
__device__ __forceinline__ int64_t rt_col_wg(const rt_lay &a, int64_t j,
                                             uint32_t wg)
{
    if (!a.ts)
        return j + a.j0;
#ifndef RT_COL_DIVIDES /* (-DRT_COL_DIVIDES: the dividing form, for A/B runs) */
    if (a.wgs)
        return j + a.j0 + (int64_t)rt_wg_block(a, wg) * (a.ts - a.bs);
#endif
    return rt_block_col(a.bs, a.ts, j + a.j0);
}

/*
 * The order in which the workgroups of a generated batch take its rays.  A
 * batch of `nf` bundles over the SAME pupil points reads every point nf times,
 * `per` workgroups (= npupil rays) apart.  While the points (16 B each) fit
 * the 256 MB Infinity Cache the later bundles find them there; above it every
 * bundle fetches them from HBM among the saturated stores: C5 on one GPU, 5
 * bundles over 2*10^7 points = 320 MB, 1.04-1.05 ms per 10^7 rays against
 * 1.006 with 160 MB (profiles/r05_final/pupil_points_sweep.jsonl).  So large
 * pupils are taken in TURNS of `turn` workgroups: turn 0 of bundle 0, turn 0
 * of bundle 1, ..., then turn 1 of every bundle: a turn's points are read
 * again while they are still cached, and the stores go on in runs of
 * turn * 256 consecutive rays.  Returns the workgroup's index in ray order
 * (wave-uniform; workgroups of padding beyond nf * per keep theirs); where
 * the results land does not change.
 */
struct rt_gen_order {
    uint32_t turn; /* workgroups per turn; 0: in ray order */
    uint32_t per;  /* workgroups per bundle */
    uint32_t nf;   /* bundles */
};

__host__ __device__ __forceinline__ uint32_t rt_gen_wg(const rt_gen_order &o,
                                                       uint32_t wg)
{
    if (!o.turn || wg >= o.per * o.nf)
        return wg;
    const uint32_t whole = o.per / o.turn; /* complete turns */
    const uint32_t lap = o.turn * o.nf;    /* workgroups of one turn of all */
    if (wg < whole * lap) {
        const uint32_t q = wg / lap, r = wg - q * lap;
        const uint32_t f = r / o.turn, t = r - f * o.turn;
        return f * o.per + q * o.turn + t;
    }
    const uint32_t last = o.per - whole * o.turn; /* > 0 here */
    const uint32_t r = wg - whole * lap;
    const uint32_t f = r / last, t = r - f * last;
    return f * o.per + whole * o.turn + t;
}

#endif /* RT_LAY_H */
