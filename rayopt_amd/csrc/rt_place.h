/*
 * rt_place.h -- WHERE the result arrays live in HBM.
 *
 * A trace writes 7-10 row streams per element at once (C3: 84 streams of
 * 40-80 MB).  The speed of that store pattern comes in two levels on MI355X
 * -- 6.7-7.0 TB/s or 5.6-5.9 -- and the trace follows it (C3 1.02 or 1.2 ms,
 * C2 0.207 or 0.259).  What decides the level (round 5, four boxes;
 * profiles/r05_probes/README.md):
 *
 *   * NOT which physical memory: the same ten 1 GiB pieces (hipMemCreate),
 *     in the same order, run the pattern at 1.157 ms behind one virtual
 *     address range and at 1.008 ms behind another (map_lab); 120
 *     arrangements of pieces drawn from all over the device memory -- one
 *     window, two or three far-apart windows, spread evenly -- are all alike
 *     inside one state of the range they are mapped into (arrange_lab
 *     "regions").  The "memory classes" of round 4 (pieces classified by a
 *     pair test, kept in an even mix) exist in that pair test -- short rows
 *     at EQUAL offsets in two pieces -- and nowhere in a real layout; the
 *     classification is gone;
 *   * NOT time, clocks or temperature: both levels are stable for 45 s side
 *     by side in one process, through idle gaps of 0.03-3 s (state_lab);
 *   * the ADDRESS RANGE the memory is mapped behind: pieces mapped ONCE into
 *     a reservation of their own are at the fast level in 12 of 12 cases and
 *     stay there; the same pieces behind a range that other mappings have
 *     used before run at either level, reproducibly per range (va_lab: two
 *     scans of one big reservation agree offset by offset); a plain hipMalloc
 *     is at the slow level (9 of 9 on four boxes; round 4: 5 of 8, 6 of 6).
 *     The mechanism is below what user space can see (page-table pages live
 *     in device memory too); the engine does not rely on an explanation.
 *
 * So large arrays are not hipMalloc'ed.  rt_place_alloc() creates pieces of
 * device memory and maps them behind one contiguous range -- an ordinary
 * device pointer to the rest of the engine -- and rt_place_tune(), once
 * rt_reserve knows the layout, MEASURES the batch's own store pattern over
 * the arrays; if that is below the fast level the same pieces are mapped
 * behind another fresh range and measured again (at most RT_PLACE_TRIES
 * ranges; the rejected reservations are held until the choice is made, so
 * that the allocator cannot hand them out again), and the best range stays.
 * 1-5 ms per range once per allocation (10^7 rays); hipMalloc if anything
 * fails: the placement is a matter of speed, never of results.
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

/* up to 1.5 GiB: hipMalloc (kernels that short live on their launch ramp) */
#define RT_PLACE_MIN_BYTES (((size_t)3 << 29) + 1)
/* the store pattern at or above this: the fast level (7.0 / 6.3 TB/s seen;
 * the slow one is 5.6-5.9), four workgroups per CU; below it two */
#define RT_PLACE_FAST_GBPS 6150.
/* two measured ranges this far apart: both levels have been seen */
#define RT_PLACE_GAP 1.07

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
    /* pieces of 1 GiB (RT_MI355_PIECE_MIB: another size, for measurements);
     * arrays below 3 GiB: 512 MiB, so that no more than a piece is wasted */
    size_t piece = (size_t)1 << 30;
    {
        const char *e = getenv("RT_MI355_PIECE_MIB");
        const long mib = e ? atol(e) : 0;
        if (mib >= 64 && mib <= 65536)
            piece = (size_t)mib << 20;
    }
    if (bytes < 3 * piece)
        piece >>= 1;
    const int need = (int)((bytes + piece - 1) / piece);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = c->device;
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)calloc(need, sizeof *h);
    int made = 0;
    hipError_t e = h ? hipSuccess : hipErrorOutOfMemory;
    for (; e == hipSuccess && made < need; ++made)
        e = hipMemCreate(&h[made], piece, &prop, 0);
    if (e != hipSuccess)
        --made; /* the failed one holds nothing */
    P.handles = h;
    P.n = e == hipSuccess ? need : (made > 0 ? made : 0);
    P.piece = piece;
    P.bytes = (size_t)need * piece;
    void *base = NULL;
    if (e == hipSuccess)
        e = rt_place_map(c, &P, &base);
    if (e != hipSuccess) {
        /* not this way (no virtual memory management, or the device is
         * full): give everything back, allocate plainly */
        (void)hipGetLastError();
        rt_place_release(&P);
        return hipMalloc(out, bytes);
    }
    P.base = base;
    P.created = need;
    P.tries = 1;
    P.search_ms = (float)(rt_place_now_ms() - t_start);
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
 * Measure, and while the arrays are at the slow level map the same pieces
 * behind other fresh ranges: the best one stays (ctx->d_buf follows).  What
 * decides between four and two workgroups per CU (rt_resident_lds) is this
 * measurement.  Batches whose pattern is too short to tell the levels apart
 * (< 0.5 GB written) keep their first range.
 */
static void rt_place_tune(rt_ctx *c, int L, long long ld)
{
    rt_place &P = c->place;
    /* (a pattern too short to measure leaves what an earlier layout found
     * out about the range the arrays live behind) */
    if (!P.base || L < 2 || 56. * (L - 1) * (double)ld < 5e8)
        return;
    P.store_gbps = 0.f;
    P.fast = 0;
    P.kept = 0;
    P.tune_ms = 0.f;
    for (int k = 0; k < RT_PLACE_TRIES; ++k)
        P.gbps[k] = 0.f;
    const double t_start = rt_place_now_ms();
    /* arrays of tens of GB: fewer ranges (a measurement writes them once) */
    const int tries = P.bytes > ((size_t)48 << 30) ? 2
                      : P.bytes > ((size_t)16 << 30) ? 3 : RT_PLACE_TRIES;
    void *range[RT_PLACE_TRIES] = {P.base};
    int n = 1, best = 0;
    P.gbps[0] = rt_place_measure(c, L, ld);
    while (n < tries && P.gbps[0] > 0.f) {
        float lo = P.gbps[0], hi = P.gbps[0];
        for (int k = 1; k < n; ++k) {
            lo = P.gbps[k] < lo ? P.gbps[k] : lo;
            hi = P.gbps[k] > hi ? P.gbps[k] : hi;
        }
        if (hi >= RT_PLACE_FAST_GBPS || hi >= RT_PLACE_GAP * lo)
            break; /* at the fast level, or both levels seen */
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
    P.fast = P.store_gbps >= RT_PLACE_FAST_GBPS;
    P.tune_ms = (float)(rt_place_now_ms() - t_start);
}

#endif /* RT_PLACE_H */
