/*
 * rt_place.h -- WHERE the result arrays live in HBM.
 *
 * A trace writes 7-10 row streams per element at once (C3: 84 streams of
 * 40-80 MB).  The speed of that store pattern is not one number on MI355X:
 * 7.0 / 6.7 / 6.2 / 5.65 TB/s have all been measured for the same pattern on
 * the same box, and the trace follows it (C3, one box, one process: 1.14 ms
 * behind a pattern of 6.97 TB/s, 1.16 at 6.86, 1.22 at 6.7, 1.27-1.29 at
 * 6.2-6.3, 1.28 in a plain hipMalloc).  Two things decide it, as far as four
 * rounds of measurements can tell (profiles/r04_probes/README.md,
 * profiles/r05_probes/README.md):
 *
 *   * WHICH pieces of device memory lie behind the arrays (round 4): 1 GiB
 *     pieces (hipMemCreate) fall into classes -- three on the boxes seen, in
 *     runs of 1-26 consecutively created pieces; a pair of pieces of one
 *     class runs a short-row store pattern at the slow level, a pair across
 *     classes at the fast one.  Arrays built from consecutive pieces as the
 *     driver hands them out, or a plain hipMalloc, sit at 6.2-6.7 / 5.65-6.0;
 *     arrays built from an even MIX of classes at 6.9-7.0 (round 5, two
 *     builds of this library alternating in one process, ten contexts each:
 *     profiles/r05_probes/s10_ab_r04_vs_pieces_mapped_once_and_hipmalloc.jsonl);
 *   * WHERE they are mapped (round 5): the same ten pieces in the same order
 *     run the pattern at 1.157 ms behind one virtual address range and at
 *     1.008 ms behind another (map_lab, va_lab: reproducible per range, two
 *     scans agree offset by offset); the level is stable over 45 s and across
 *     idle gaps (state_lab).  This is what made C2 bimodal in round 4 -- 0.207
 *     or 0.259 ms with the SAME class mix.  The mechanism is below what user
 *     space can see (page tables live in device memory too); the engine does
 *     not rely on an explanation, it measures.
 *
 * So large arrays are not hipMalloc'ed.  rt_place_alloc() creates pieces of
 * device memory, finds the class of each with a pair test (42 short row
 * streams in the piece, 42 in a representative of a known class: slow =
 * same class), keeps a balanced mix, releases the rest and maps the kept
 * pieces, classes interleaved, behind ONE contiguous range of its own -- what
 * the rest of the engine sees is an ordinary device pointer.  rt_place_tune(),
 * once rt_reserve knows the layout, then writes the batch's OWN store pattern
 * over the arrays and times it; below RT_PLACE_GOOD_GBPS the same pieces are
 * mapped behind a second fresh range and measured again (the first
 * reservation is held meanwhile, so that the allocator cannot hand it out
 * again), and rt_place_settle() goes on to ANOTHER set of pieces while the
 * first is held -- sets of one process come out anywhere between 6300 and
 * 7050 GB/s -- at most three sets; the best stays.  15-60 ms once per
 * allocation (more where the driver is slow to hand out memory it has just
 * got back).  Anything that fails on the way
 * (no virtual memory management, out of memory for the surplus) falls back
 * to hipMalloc: the placement is a matter of speed, never of results.
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
/* the batch's own store pattern (rt_place_tune): at or above GOOD no other
 * range is tried; below FAST the arrays behave like one class whatever the
 * pair tests said (four workgroups per CU then lose to two); two ranges this
 * far apart (GAP): both ends of what this memory does have been seen.
 * Sets of pieces come out anywhere between 6300 and 7050 GB/s in one process
 * (a lottery the classes narrow but do not end); GOOD is where the search
 * stops paying: bundles with per-ray launch directions (reads among the
 * saturated writes) trace at 1.077 ms behind 6950 GB/s, 1.10 behind 6820,
 * 1.15 behind 6770, C2 at 0.208 behind 6950 and 0.235-0.241 behind 6300 */
#define RT_PLACE_GOOD_GBPS 6800.
#define RT_PLACE_FAST_GBPS 5950.
#define RT_PLACE_GAP 1.07

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

/* the pieces of `p` behind a fresh address range */
static hipError_t rt_place_map(rt_ctx *c, rt_place *p, void **out)
{
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device; /* (this device only: a pointer handed out by
                                    rt_device_ptr is not a peer / IPC pointer) */
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const hipMemGenericAllocationHandle_t *h =
        (const hipMemGenericAllocationHandle_t *)p->handles;
    void *base = NULL;
    hipError_t e = hipMemAddressReserve(&base, p->bytes, p->piece, NULL, 0);
    int nm = 0;
    for (; e == hipSuccess && nm < p->n; ++nm)
        e = hipMemMap((char *)base + (size_t)nm * p->piece, p->piece, 0, h[nm],
                      0);
    if (e == hipSuccess)
        e = hipMemSetAccess(base, p->bytes, &acc, 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int k = 0; k < nm; ++k) /* (the last one may not have taken) */
            (void)hipMemUnmap((char *)base + (size_t)k * p->piece, p->piece);
        (void)hipGetLastError();
        if (base)
            (void)hipMemAddressFree(base, p->bytes);
        return e;
    }
    *out = base;
    return hipSuccess;
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
    /* what the search may hold beyond the `need` pieces it keeps -- surplus
     * pieces and ballast -- stays below half of the memory that is free
     * now: other contexts and processes allocate from the same device */
    size_t extra = 0, budget = 0;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess)
            (void)hipGetLastError();
        const size_t keep_b = (size_t)need * piece;
        budget = free_b / 2 > keep_b ? (free_b / 2 - keep_b) / 1 : 0;
        if (free_b < keep_b + 2 * piece) /* no room to choose from */
            return hipMalloc(out, bytes);
    }
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
        if (made >= need && extra + piece > budget)
            break; /* the surplus has reached its share of the free memory */
        if (hipMemCreate(&h[k], piece, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break; /* the device is full: what exists must do */
        }
        ++made;
        if (made > need)
            extra += piece;
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
            if (extra + hop > budget) {
                /* (no room left to hop in) */
            } else if (hipMemCreate(&ballast[nballast], hop, &prop, 0) ==
                       hipSuccess) {
                ++nballast;
                extra += hop;
            } else {
                (void)hipGetLastError(); /* the device is full: no hopping */
            }
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
    /* the scratch range goes back first: the final range then begins where
     * the pair tests ran (measured against a range reserved while the
     * scratch was still held, two builds alternating in one process: the
     * store pattern 6750-6865 GB/s here, 6555-6676 there; where even this
     * range is slow rt_place_tune tries others) */
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
    P.fast = P.mixed; /* (until the pattern itself has been measured) */
    P.tries = 1;
    const double t_end = rt_place_now_ms();
    P.search_ms = (float)(t_end - t_start);
    P.ballast_ms = (float)t_ballast;
    P.pieces_ms = (float)(t_found - t_start - t_created);
    P.remap_ms = (float)(t_end - t_found - (t_ballast - t_created));
    *out = base;
    return hipSuccess;
}


/*
 * The trace's own store pattern -- y0 y1 y2 u0 u1 u2 t of every element, one
 * ray per lane -- over the arrays as they are laid out.  Rows are
 * overwritten: called from rt_reserve, before anything lives in them.
 */
__global__ __launch_bounds__(256) void rt_place_rows_kernel(rt_lay a, int L,
                                                            long long n)
{
    const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n)
        return;
    const double v = 1e-9 * (double)j;
    const long long r = rt_col_wg(a, j, blockIdx.x);
    for (int s = 1; s < L; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            __builtin_nontemporal_store(v + c, &a.Y[(long long)s * a.ss + c * a.cs + r]);
            __builtin_nontemporal_store(v - c, &a.U[(long long)s * a.ss + c * a.cs + r]);
        }
        __builtin_nontemporal_store(v, &a.T[(long long)s * a.ssT + r]);
    }
}

/* GB/s of the pattern over the arrays as ctx->d_buf now maps them (0: could
 * not be measured) */
static float rt_place_measure(rt_ctx *c, int L, long long ld)
{
    const double bytes = 56. * (L - 1) * (double)ld;
    const rt_lay lay = rt_layout(c);
    const unsigned grid = (unsigned)((ld + 255) / 256);
    /* ~4 ms of launches, at least three: short kernels (C2: 0.2 ms) are
     * timed over more of them */
    int reps = (int)(4e-3 / (bytes / 6.5e12)) + 1;
    reps = reps < 3 ? 3 : (reps > 24 ? 24 : reps);
    hipLaunchKernelGGL(rt_place_rows_kernel, dim3(grid), dim3(256), 32768,
                       c->stream, lay, L, ld);
    if (hipEventRecord(c->k0, c->stream) != hipSuccess)
        return 0.f;
    for (int k = 0; k < reps; ++k)
        hipLaunchKernelGGL(rt_place_rows_kernel, dim3(grid), dim3(256), 32768,
                           c->stream, lay, L, ld);
    float ms = 0.f, gbps = 0.f;
    if (hipEventRecord(c->k1, c->stream) == hipSuccess &&
        hipEventSynchronize(c->k1) == hipSuccess &&
        hipEventElapsedTime(&ms, c->k0, c->k1) == hipSuccess && ms > 0.f)
        gbps = (float)(reps * bytes / (ms * 1e-3) / 1e9);
    (void)hipGetLastError();
    return gbps;
}

/*
 * Measure, and while the pattern is below RT_PLACE_GOOD_GBPS map the same
 * pieces behind other fresh ranges: the best one stays (ctx->d_buf follows).
 * What decides between four and two workgroups per CU (rt_resident_lds) is
 * this measurement, not the classes.  Batches whose pattern is too short to
 * tell anything (< 0.5 GB written) keep their first range.
 */
static void rt_place_tune(rt_ctx *c, int L, long long ld)
{
    rt_place &P = c->place;
    /* (a pattern too short to measure leaves what the classes said, or
     * what an earlier layout found out about the range) */
    if (!P.base || L < 2 || 56. * (L - 1) * (double)ld < 5e8)
        return;
    P.store_gbps = 0.f;
    P.kept = 0;
    P.tune_ms = 0.f;
    for (int k = 0; k < RT_PLACE_TRIES; ++k)
        P.gbps[k] = 0.f;
    const double t_start = rt_place_now_ms();
    /* two ranges at most: in this engine's allocations the ranges of one set
     * of pieces have turned out within 2 % of each other (it is the pieces
     * that the batches below the fast level want exchanged: rt_place_settle) */
    const int tries = 2;
    void *range[RT_PLACE_TRIES] = {P.base};
    int n = 1, best = 0;
    P.gbps[0] = rt_place_measure(c, L, ld);
    while (n < tries && P.gbps[0] > 0.f) {
        float lo = P.gbps[0], hi = P.gbps[0];
        for (int k = 1; k < n; ++k) {
            lo = P.gbps[k] < lo ? P.gbps[k] : lo;
            hi = P.gbps[k] > hi ? P.gbps[k] : hi;
        }
        if (hi >= c->opt_place_good || hi >= RT_PLACE_GAP * lo)
            break; /* as good as it gets, or both ends seen */
        /* the same pieces behind another range; the ranges tried so far
         * stay reserved so that the next one is a new one */
        if (hipStreamSynchronize(c->stream) != hipSuccess ||
            hipMemUnmap(range[n - 1], P.bytes) != hipSuccess)
            break; /* (cannot happen; the arrays stay where they are) */
        void *next = NULL;
        if (rt_place_map(c, &P, &next) != hipSuccess) {
            (void)hipGetLastError();
            void *again = NULL; /* back behind a range that worked */
            (void)hipMemAddressFree(range[n - 1], P.bytes);
            if (rt_place_map(c, &P, &again) == hipSuccess)
                range[n - 1] = again;
            else
                range[n - 1] = NULL;
            break;
        }
        range[n] = next;
        P.base = next;
        c->d_buf = (double *)next;
        P.gbps[n] = rt_place_measure(c, L, ld);
        ++n;
    }
    for (int k = 1; k < n; ++k)
        if (P.gbps[k] > P.gbps[best])
            best = k;
    if (range[n - 1] == NULL) {
        /* lost the mapping on the way (device out of address space?): the
         * caller sees a failed allocation */
        P.base = NULL;
        c->d_buf = NULL;
    } else if (best != n - 1 && range[best]) {
        /* the winner is an earlier range: back behind it.  Its reservation
         * is still held; the pieces go where they were (va_lab: a range
         * keeps its level when the same pieces return to it) */
        const hipMemGenericAllocationHandle_t *h =
            (const hipMemGenericAllocationHandle_t *)P.handles;
        hipMemAccessDesc acc = {};
        acc.location.type = hipMemLocationTypeDevice;
        acc.location.id = c->device;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        hipError_t e = hipStreamSynchronize(c->stream);
        if (e == hipSuccess)
            e = hipMemUnmap(range[n - 1], P.bytes);
        int nm = 0;
        for (; e == hipSuccess && nm < P.n; ++nm)
            e = hipMemMap((char *)range[best] + (size_t)nm * P.piece, P.piece,
                          0, h[nm], 0);
        if (e == hipSuccess)
            e = hipMemSetAccess(range[best], P.bytes, &acc, 1);
        if (e == hipSuccess) {
            P.base = range[best];
            c->d_buf = (double *)range[best];
        } else { /* stay behind the last range */
            (void)hipGetLastError();
            for (int k = 0; k < nm; ++k)
                (void)hipMemUnmap((char *)range[best] + (size_t)k * P.piece,
                                  P.piece);
            nm = 0;
            for (e = hipSuccess; e == hipSuccess && nm < P.n; ++nm)
                e = hipMemMap((char *)range[n - 1] + (size_t)nm * P.piece,
                              P.piece, 0, h[nm], 0);
            if (e == hipSuccess)
                e = hipMemSetAccess(range[n - 1], P.bytes, &acc, 1);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                P.base = NULL;
                c->d_buf = NULL;
            }
            best = n - 1;
        }
    }
    for (int k = 0; k < n; ++k)
        if (range[k] && range[k] != P.base)
            (void)hipMemAddressFree(range[k], P.bytes);
    P.tries = n;
    P.kept = best;
    P.store_gbps = P.gbps[best];
    if (P.store_gbps > 0.f) /* whatever the classes said */
        P.fast = P.store_gbps >= RT_PLACE_FAST_GBPS;
    P.tune_ms = (float)(rt_place_now_ms() - t_start);
}

/*
 * rt_place_tune, and while the arrays stay below RT_PLACE_GOOD_GBPS ANOTHER
 * set of pieces: the current one is held (so that the new pieces come from
 * elsewhere in the device memory), classified, mapped and measured like the
 * first, and the better set stays.  At most RT_PLACE_PICKS sets, arrays up
 * to 16 GiB (C2: 0.2414 ms behind pieces whose four ranges all ran the
 * pattern at 5.45-5.5 TB/s, 0.207-0.218 behind others).  ctx->d_buf follows.
 */
#define RT_PLACE_PICKS 3
static hipError_t rt_place_alloc(rt_ctx *c, void **out, size_t bytes);

static void rt_place_settle(rt_ctx *c, int L, long long ld, size_t bytes)
{
    if (!c->place.base || L < 2 || 56. * (L - 1) * (double)ld < 5e8)
        return; /* (a pattern too short to tell anything: what an earlier
                   layout found out about these arrays stays) */
    rt_place_tune(c, L, ld);
    int picks = 1;
    float seen[RT_PLACE_PICKS] = {c->place.store_gbps};
    while (picks < RT_PLACE_PICKS && c->d_buf && c->place.base &&
           c->place.store_gbps > 0.f &&
           c->place.store_gbps < c->opt_place_good &&
           c->place.bytes <= ((size_t)16 << 30)) {
        const rt_place held = c->place; /* pieces and range stay alive */
        double *const held_buf = c->d_buf;
        void *nb = NULL;
        const hipError_t e = rt_place_alloc(c, &nb, bytes);
        if (e != hipSuccess || !c->place.base) {
            /* no second set to be had (the fallback's hipMalloc is not one) */
            (void)hipGetLastError();
            if (e == hipSuccess && nb)
                (void)hipFree(nb);
            c->place = held;
            c->d_buf = held_buf;
            break;
        }
        c->d_buf = (double *)nb;
        rt_place_tune(c, L, ld);
        seen[picks++] = c->place.store_gbps;
        if (!c->d_buf || c->place.store_gbps <= held.store_gbps) {
            if (c->place.base || c->place.handles)
                rt_place_release(&c->place); /* no better: back to the held */
            c->place = held;
            c->d_buf = held_buf;
        } else {
            rt_place tmp = held;
            rt_place_release(&tmp);
        }
    }
    c->place.picks = picks;
    for (int k = 0; k < RT_PLACE_PICKS; ++k)
        c->place.pick_gbps[k] = k < picks ? seen[k] : 0.f;
}

#endif /* RT_PLACE_H */
