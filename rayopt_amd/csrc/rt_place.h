/*
 * rt_place.h -- WHERE the result arrays live in HBM (round 4).
 *
 * A trace writes 7-10 row streams per element at once (C3: 84 streams of
 * 80 MB, 80 MB apart).  Measured on MI355X (profiles/r04_probes/README.md):
 *
 *   * the speed of that store pattern is a property of the PHYSICAL memory
 *     behind the arrays and comes in levels: 0.96 / 1.07 / 1.19 ms per 10^7
 *     rays (7.0 / 6.3 / 5.65 TB/s), and the trace follows it: 1.12 / 1.19 /
 *     1.35 ms at four workgroups per CU.  The counters put the difference at
 *     the DRAM side (TCC_EA0_WRREQ_DRAM_CREDIT_STALL x7 in a slow allocation,
 *     address-translation misses equal);
 *   * 1 GiB pieces of device memory (hipMemCreate) fall into CLASSES -- three
 *     on the boxes seen, in runs of 2-14 consecutively created pieces: all
 *     84 streams inside pieces of ONE class run at the slow level (a single
 *     piece, a plain hipMalloc of 10 GB and a 16 GiB buddy block are that
 *     case), streams dealt over pieces of TWO OR THREE classes at the fast
 *     one; which pieces and in which order does not matter, only the mix.
 *     (Consistent with the three stack IDs of a 12-high HBM3E stack being
 *     selected by high physical address bits: banks of different stack IDs
 *     do not conflict.  User space cannot see physical addresses; the class
 *     of a piece is MEASURED.)
 *
 * So large arrays are not hipMalloc'ed.  rt_place_alloc() creates pieces of
 * device memory, finds the class of each with a pair test (42 short row
 * streams in the piece, 42 in a representative of a known class: slow =
 * same class), keeps a balanced mix, releases the rest and maps the kept
 * pieces, classes interleaved, behind ONE contiguous address range -- what
 * the rest of the engine sees is an ordinary device pointer.  ~1 ms per
 * piece, once per allocation.  Anything that fails on the way (no virtual
 * memory management, out of memory for the surplus) falls back to hipMalloc:
 * the placement is a matter of speed, never of results.
 */
#ifndef RT_PLACE_H
#define RT_PLACE_H

#include "rt_ctx.h"
#include <chrono>

static inline double rt_place_now_ms(void)
{
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define RT_PLACE_ROWS 84         /* 12 elements x (y0 y1 y2 u0 u1 u2 t) */
/* up to 1.5 GiB: hipMalloc.  Above it there are at least four pieces of
 * 512 MiB, two per class: with three the rows of Y, U and T fall on the
 * classes in lumps ([2, 1]: two thirds of the streams in one class) and the
 * trace is as often slower as faster (profiles/r04_probes/session26) */
#define RT_PLACE_MIN_BYTES (((size_t)3 << 29) + 1)
#define RT_PLACE_SAME 0.91f      /* pair / self time above this: same class */

struct rt_place_rows {
    double *row[RT_PLACE_ROWS];
};

/* the trace kernel's store pattern, every row stream through a pointer */
__global__ __launch_bounds__(256) void rt_place_pair_kernel(rt_place_rows tb,
                                                             long long n)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= n)
        return;
    const double a = 1e-9 * (double)r;
    for (int s = 0; s < RT_PLACE_ROWS / 7; ++s) {
#pragma unroll
        for (int j = 0; j < 7; ++j)
            tb.row[s * 7 + j][r] = a + j;
    }
}

/* rows 0..41 in piece a, 42..83 in piece b (a == b: all 84 in it) */
static hipError_t rt_place_time(rt_ctx *c, double *a, double *b, long long n,
                                float *ms)
{
    rt_place_rows tb;
    for (int s = 0; s < RT_PLACE_ROWS; ++s) {
        const int half = RT_PLACE_ROWS / 2;
        if (a == b)
            tb.row[s] = a + (long long)s * n;
        else
            tb.row[s] = (s < half ? a : b) + (long long)(s % half) * n;
    }
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(rt_place_pair_kernel, dim3(grid), dim3(256), 32768,
                       c->stream, tb, n);
    hipError_t e = hipEventRecord(c->k0, c->stream);
    for (int k = 0; k < 3 && e == hipSuccess; ++k)
        hipLaunchKernelGGL(rt_place_pair_kernel, dim3(grid), dim3(256), 32768,
                           c->stream, tb, n);
    if (e == hipSuccess)
        e = hipEventRecord(c->k1, c->stream);
    if (e == hipSuccess)
        e = hipEventSynchronize(c->k1);
    if (e == hipSuccess)
        e = hipEventElapsedTime(ms, c->k0, c->k1);
    if (e == hipSuccess)
        e = hipGetLastError();
    return e;
}

static void rt_place_release(rt_place *p)
{
    if (p->base) {
        (void)hipMemUnmap(p->base, p->bytes);
        (void)hipMemAddressFree(p->base, p->bytes);
    }
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)p->handles;
    for (int k = 0; k < p->n; ++k)
        (void)hipMemRelease(h[k]);
    free(p->handles);
    memset(p, 0, sizeof *p);
}

/* frees `ptr` whether it came from rt_place_alloc's mapping or hipMalloc */
static hipError_t rt_place_free(rt_ctx *c, void *ptr)
{
    if (ptr && ptr == c->place.base) {
        rt_place_release(&c->place);
        return hipSuccess;
    }
    return ptr ? hipFree(ptr) : hipSuccess;
}

static hipError_t rt_place_alloc(rt_ctx *c, void **out, size_t bytes)
{
    rt_place &P = c->place;
    memset(&P, 0, sizeof P);
    if (!c->opt_place || bytes < RT_PLACE_MIN_BYTES)
        return hipMalloc(out, bytes);
    const double t_start = rt_place_now_ms();
    double t_ballast = 0.;

    /* pieces of 1 GiB (RT_MI355_PIECE_MIB: another size, for measurements);
     * arrays below 3 GiB: pieces of 512 MiB */
    size_t piece = (size_t)1 << 30;
    {
        const char *e = getenv("RT_MI355_PIECE_MIB");
        const long mib = e ? atol(e) : 0;
        if (mib >= 512 && mib <= 65536)
            piece = (size_t)mib << 20;
    }
    if (bytes < 3 * piece)
        piece >>= 1; /* 512 MiB: a power of two, i.e. ONE block of the
                        device's buddy allocator -- a 768 MiB piece is two
                        blocks that may lie in two classes, and its "one
                        piece" time is then already the fast one */
    const int need = (int)((bytes + piece - 1) / piece);
    size_t align = piece & (~piece + 1); /* largest power of two dividing it */
    const int cap = need + 24; /* pieces created and classified at most */
    /* Pieces come in runs of one class (2-14 seen, 26+ on one box): once a
     * class is oversupplied the search HOPS -- a block of ballast is created
     * and held, unclassified, so that the next piece lies further on in the
     * device memory -- until another class turns up.  Ballast and surplus
     * pieces go back to the device before rt_place_alloc returns. */
    const int max_ballast = 24;
    hipMemGenericAllocationHandle_t ballast[24];
    int nballast = 0, hops_in_a_row = 0;
    /* short rows: 84 of them fit one piece */
    const long long nprobe =
        (long long)(piece / sizeof(double) / RT_PLACE_ROWS) / 256 * 256;

    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;

    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)calloc(cap, sizeof *h);
    unsigned char *cls = (unsigned char *)calloc(cap, 1);
    void *scratch = NULL; /* every created piece at scratch + k * piece */
    int made = 0, mapped = 0, nclass = 0, rep[RT_PLACE_CLASSES];
    int count[RT_PLACE_CLASSES] = {0};
    float self_ms = 0.f, cross_ms = 0.f;
    hipError_t e = h && cls ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess)
        e = hipMemAddressReserve(&scratch, (size_t)cap * piece, align, NULL, 0);
    bool enough = false;
    while (e == hipSuccess && made < cap && !enough) {
        const int k = made;
        if (hipMemCreate(&h[k], piece, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break; /* the device is full: what exists must do */
        }
        ++made;
        double *pk = (double *)((char *)scratch + (size_t)k * piece);
        e = hipMemMap(pk, piece, 0, h[k], 0);
        if (e == hipSuccess) {
            ++mapped;
            e = hipMemSetAccess(pk, piece, &acc, 1);
        }
        if (e != hipSuccess)
            break;
        float ms = 0.f;
        if (k == 0) {
            /* the slow level: all rows in one piece -- repeated until two
             * measurements agree to 2 % (a device coming out of idle) */
            e = rt_place_time(c, pk, pk, nprobe, &ms);
            for (int w = 0; w < 12 && e == hipSuccess; ++w) {
                e = rt_place_time(c, pk, pk, nprobe, &self_ms);
                const bool steady = fabsf(self_ms - ms) <= .02f * self_ms;
                ms = self_ms;
                if (steady)
                    break;
            }
            if (e != hipSuccess)
                break;
            cls[0] = 0;
            rep[0] = 0;
            nclass = 1;
            count[0] = 1;
        } else {
            /* pieces come in runs of one class: the previous one's first */
            int order[RT_PLACE_CLASSES], no = 0;
            order[no++] = cls[k - 1];
            for (int q = 0; q < nclass; ++q)
                if (q != cls[k - 1])
                    order[no++] = q;
            int found = -1;
            for (int q = 0; q < no && found < 0; ++q) {
                double *pr = (double *)((char *)scratch +
                                        (size_t)rep[order[q]] * piece);
                e = rt_place_time(c, pk, pr, nprobe, &ms);
                if (e != hipSuccess)
                    break;
                if (ms > RT_PLACE_SAME * self_ms)
                    found = order[q];
                else
                    cross_ms = ms;
            }
            if (e != hipSuccess)
                break;
            if (found < 0) {
                if (nclass < RT_PLACE_CLASSES) {
                    found = nclass++;
                    rep[found] = k;
                } else {
                    found = RT_PLACE_CLASSES - 1; /* more kinds than room */
                }
            }
            cls[k] = (unsigned char)found;
            ++count[found];
        }
        if (count[cls[k]] > (need + 1) / 2 && made >= (need + 1) / 2 + 1 &&
            nballast < max_ballast) {
            const size_t hop = (size_t)(hops_in_a_row < 3 ? 4 : 8) << 30;
            const double tb = rt_place_now_ms();
            if (hipMemCreate(&ballast[nballast], hop, &prop, 0) == hipSuccess)
                ++nballast;
            else
                (void)hipGetLastError(); /* the device is full: no hopping */
            t_ballast += rt_place_now_ms() - tb;
            ++hops_in_a_row;
        } else {
            hops_in_a_row = 0;
        }
        /* enough when `need` pieces can be picked with no class holding
         * more than half of them (two classes evenly mixed run at 0.98 of
         * the three-class time: not worth a dozen more pieces) */
        if (made >= need && nclass >= 2) {
            int can = 0;
            for (int q = 0; q < nclass; ++q)
                can += count[q] < (need + 1) / 2 ? count[q] : (need + 1) / 2;
            enough = can >= need;
        }
    }
    const double t_found = rt_place_now_ms(), t_created = t_ballast;
    for (int b = 0; b < nballast; ++b)
        (void)hipMemRelease(ballast[b]);
    t_ballast += rt_place_now_ms() - t_found;
    if (e != hipSuccess || made < need) {
        /* not this way: give everything back, allocate plainly */
        (void)hipGetLastError();
        for (int k = 0; k < mapped; ++k)
            (void)hipMemUnmap((char *)scratch + (size_t)k * piece, piece);
        for (int k = 0; k < made; ++k)
            (void)hipMemRelease(h[k]);
        if (scratch)
            (void)hipMemAddressFree(scratch, (size_t)cap * piece);
        free(h);
        free(cls);
        return hipMalloc(out, bytes);
    }
    /* pick `need` pieces round-robin over the classes (an even mix, as far
     * as the counts allow), in that order along the address range */
    int *pick = (int *)calloc(need, sizeof(int));
    hipMemGenericAllocationHandle_t *kept =
        (hipMemGenericAllocationHandle_t *)calloc(need, sizeof *kept);
    int next[RT_PLACE_CLASSES] = {0}, taken = 0, q = 0, idle = 0;
    int used[RT_PLACE_CLASSES] = {0};
    while (pick && taken < need && idle < nclass) {
        int k = next[q];
        while (k < made && cls[k] != q)
            ++k;
        if (k < made) {
            pick[taken++] = k;
            next[q] = k + 1;
            ++used[q];
            idle = 0;
        } else {
            next[q] = made;
            ++idle;
        }
        q = (q + 1) % nclass;
    }
    for (int k = 0; k < made; ++k)
        (void)hipMemUnmap((char *)scratch + (size_t)k * piece, piece);
    (void)hipMemAddressFree(scratch, (size_t)cap * piece);
    void *base = NULL;
    e = pick && kept && taken == need ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess)
        e = hipMemAddressReserve(&base, (size_t)need * piece, align, NULL, 0);
    int nm = 0;
    for (; e == hipSuccess && nm < need; ++nm) {
        e = hipMemMap((char *)base + (size_t)nm * piece, piece, 0,
                      h[pick[nm]], 0);
        if (e == hipSuccess) {
            kept[nm] = h[pick[nm]];
            h[pick[nm]] = 0;
        }
    }
    if (e == hipSuccess)
        e = hipMemSetAccess(base, (size_t)need * piece, &acc, 1);
    for (int k = 0; k < made; ++k) /* the surplus */
        if (h[k])
            (void)hipMemRelease(h[k]);
    free(h);
    free(cls);
    free(pick);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int k = 0; k < nm && kept; ++k) {
            if (kept[k]) { /* mapped */
                (void)hipMemUnmap((char *)base + (size_t)k * piece, piece);
                (void)hipMemRelease(kept[k]);
            }
        }
        if (base)
            (void)hipMemAddressFree(base, (size_t)need * piece);
        free(kept);
        return hipMalloc(out, bytes);
    }
    P.base = base;
    P.bytes = (size_t)need * piece;
    P.piece = piece;
    P.n = need;
    P.handles = kept;
    P.created = made;
    P.nclass = nclass;
    for (int k = 0; k < RT_PLACE_CLASSES; ++k)
        P.count[k] = used[k];
    P.self_ms = self_ms;
    P.cross_ms = cross_ms;
    /* mixed: at least a third of the pieces lie outside the largest class */
    int largest = 0;
    for (int k = 0; k < nclass; ++k)
        largest = used[k] > largest ? used[k] : largest;
    P.mixed = nclass >= 2 && 3 * (need - largest) >= need;
    P.ballast = nballast;
    P.class_mix = P.mixed;
    const double t_end = rt_place_now_ms();
    P.search_ms = (float)(t_end - t_start);
    P.ballast_ms = (float)t_ballast;
    P.pieces_ms = (float)(t_found - t_start - t_created);
    P.remap_ms = (float)(t_end - t_found - (t_ballast - t_created));
    *out = base;
    return hipSuccess;
}

/*
 * The proof of the pudding: the trace's own store pattern -- y0 y1 y2 u0 u1
 * u2 t of every element, one ray per lane -- over the arrays as they are now
 * laid out, three launches.  Classes are a model (three on most boxes seen;
 * one box traced at the slow level in a mix that should have been fast):
 * what decides between four and two workgroups per CU is this measurement.
 * Levels of the bare pattern: 7.0 / 6.3 / 5.65 TB/s (mixed / partly mixed /
 * one class); 18 placed contexts on three boxes: 6.73-7.04.  Rows are
 * overwritten: called from rt_reserve, before anything lives in them.
 */
/* below this the arrays behave like ONE class (5.65 TB/s; four workgroups per
 * CU then lose to two); the middle level (6.3) still takes four better */
#define RT_PLACE_FAST_GBPS 5950.
#define RT_PLACE_VERIFY_BYTES ((size_t)4 << 30)

__global__ __launch_bounds__(256) void rt_place_rows_kernel(rt_lay a, int L,
                                                            long long n)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n)
        return;
    const double v = 1e-9 * (double)j;
    const long long r = rt_col(a, j);
    for (int s = 1; s < L; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            a.Y[(long long)s * a.ss + c * a.cs + r] = v + c;
            a.U[(long long)s * a.ss + c * a.cs + r] = v - c;
        }
        a.T[(long long)s * a.ssT + r] = v;
    }
}

static void rt_place_verify(rt_ctx *c, rt_lay lay, int L, long long ld)
{
    rt_place &P = c->place;
    P.store_gbps = 0.f;
    const double t_start = rt_place_now_ms();
    const size_t bytes = (size_t)56 * (L - 1) * ld;
    if (!P.base || L < 2 || bytes < RT_PLACE_VERIFY_BYTES)
        return; /* short kernels measure their own ramp, not the memory */
    const unsigned grid = (unsigned)((ld + 255) / 256);
    hipLaunchKernelGGL(rt_place_rows_kernel, dim3(grid), dim3(256), 32768,
                       c->stream, lay, L, ld);
    if (hipEventRecord(c->k0, c->stream) != hipSuccess)
        return;
    for (int k = 0; k < 3; ++k)
        hipLaunchKernelGGL(rt_place_rows_kernel, dim3(grid), dim3(256), 32768,
                           c->stream, lay, L, ld);
    float ms = 0.f;
    if (hipEventRecord(c->k1, c->stream) == hipSuccess &&
        hipEventSynchronize(c->k1) == hipSuccess &&
        hipEventElapsedTime(&ms, c->k0, c->k1) == hipSuccess && ms > 0.f) {
        P.store_gbps = (float)(3. * (double)bytes / (ms * 1e-3) / 1e9);
        if (P.store_gbps < RT_PLACE_FAST_GBPS)
            P.mixed = 0; /* whatever the classes say: two per CU here */
    }
    (void)hipGetLastError();
    P.verify_ms = (float)(rt_place_now_ms() - t_start);
}

#endif /* RT_PLACE_H */
