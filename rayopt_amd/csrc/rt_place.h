/*
 * rt_place.h -- WHERE the result arrays live in HBM.
 *
 * A trace writes 7-10 row streams per element at once (C3: 84 streams of
 * 40-80 MB).  The speed of that store pattern is not one number on MI355X --
 * 7.0 / 6.7 / 6.2 / 5.65 TB/s have all been measured for the same pattern on
 * one box -- and the trace follows it (C3, one process: 1.14 ms behind a
 * pattern of 6.97 TB/s, 1.16 at 6.86, 1.22 at 6.7, 1.27-1.29 at 6.2-6.3, 1.28
 * in a plain hipMalloc).  What decides it is WHICH pieces of device memory lie
 * behind the arrays (profiles/r04_probes/README.md,
 * profiles/r05_probes/README.md):
 *
 *   * 1 GiB pieces (hipMemCreate) fall into classes -- three on the boxes
 *     seen, in runs of 1-26 consecutively created pieces; a pair of pieces of
 *     one class runs a short-row store pattern at the slow level, a pair
 *     across classes at the fast one.  Arrays built from consecutive pieces
 *     as the driver hands them out, or a plain hipMalloc, sit at 6.2-6.7 /
 *     5.65-6.0; arrays built from an even MIX of classes higher (two builds
 *     of this library alternating in one process:
 *     profiles/r05_probes/s10_*.jsonl);
 *   * THREE classes beat two: sets of two classes 6.67-6.77 TB/s in five of
 *     five contexts, sets of three 6.91-7.02 in nine of ten (round 5), so the
 *     search goes on for a third class;
 *   * sets of equal class counts still differ (C2, five 512 MiB pieces: 5.5
 *     to 7.0 TB/s in one process -- round 4's "bimodal with the same class
 *     mix"): the batch's OWN pattern is measured over the arrays, and while
 *     it is low another set of pieces is tried.
 *
 * So large arrays are not hipMalloc'ed.  rt_place_alloc() creates pieces of
 * device memory, finds the class of each with a pair test (42 short row
 * streams in the piece, 42 in a representative of a known class: slow =
 * same class), keeps a balanced mix, releases the rest and maps the kept
 * pieces, classes interleaved, behind ONE contiguous range -- what the rest
 * of the engine sees is an ordinary device pointer.  rt_place_settle(), once
 * rt_reserve knows the layout, writes the batch's own store pattern over the
 * arrays and times it; below RT_PLACE_GOOD_GBPS it goes on to another set of
 * pieces while the first is held (at most three sets, five for small arrays
 * and where three were clearly not enough); the best stays.
 * 15-60 ms once per allocation (more where the driver is slow to hand out
 * memory it has just got back).  Anything that fails on the way (no virtual
 * memory management, out of memory for the surplus) falls back to hipMalloc:
 * the placement is a matter of speed, never of results -- provided the
 * platform's hazard below is respected (rt_place_flush, rt_place_coherent).
 */
#ifndef RT_PLACE_H
#define RT_PLACE_H

#include "rt_ctx.h"

static inline double rt_place_now_ms(void) { return rt_now_ms(); }

#define RT_PLACE_ROWS 84         /* 12 elements x (y0 y1 y2 u0 u1 u2 t) */
/* up to 1.5 GiB: hipMalloc.  Above it there are at least four pieces of
 * 512 MiB, two per class: with three the rows of Y, U and T fall on the
 * classes in lumps ([2, 1]: two thirds of the streams in one class) and the
 * trace is as often slower as faster (profiles/r04_probes/session26) */
#define RT_PLACE_MIN_BYTES (((size_t)3 << 29) + 1)
/* pair / self time above this: same class.  The pairs of 512 MiB and 1 GiB
 * pieces fall on THREE levels -- 0.74-0.81 (another class), 0.87-0.93 and
 * 0.96-1.05 (scripts/c2_lab.py, the 5 x 5 matrices of round 6) -- and only
 * the first is what the placement is after: until round 6 the mark stood at
 * 0.91, INSIDE the middle level, and pieces of one class were told apart at
 * random (a set labelled A B C A B whose B were A in fact) */
#define RT_PLACE_SAME 0.85f
/* the batch's own store pattern (rt_place_tune): at or above GOOD no other
 * range is tried; below FAST the arrays behave like one class whatever the
 * pair tests said (four workgroups per CU then lose to two).
 * Sets of pieces come out anywhere between 6300 and 7050 GB/s in one process
 * (a lottery the classes narrow but do not end); GOOD is where the search
 * stops paying: bundles with per-ray launch directions (reads among the
 * saturated writes) trace at 1.077 ms behind 6950 GB/s, 1.10 behind 6820,
 * 1.15 behind 6770, C2 at 0.208 behind 6950 and 0.235-0.241 behind 6300 */
#define RT_PLACE_GOOD_GBPS 6900.
#define RT_PLACE_FAST_GBPS 5950.

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

/*
 * The virtual-memory calls that give something back (hipMemUnmap,
 * hipMemRelease, hipMemAddressFree) have nothing to fall back to, but a
 * failure of one is not silent: it is counted per process, the first one is
 * remembered with its call site, rt_placement reports the count and
 * rt_reserve puts the text where rt_last_error finds it.
 */
static int g_place_vm_failures = 0;
static char g_place_vm_first[160] = "";

static inline void rt_place_vm(hipError_t e, const char *what)
{
    if (e == hipSuccess)
        return;
    (void)hipGetLastError();
    if (!g_place_vm_failures++)
        snprintf(g_place_vm_first, sizeof g_place_vm_first, "%s: %s", what,
                 hipGetErrorString(e));
}

static void rt_place_flush(void);

/*
 * Handing a virtual address range back.  (Round 6 tried never to do that --
 * an address that is not reused cannot be reached through a stale
 * translation -- and ran out of address space after about 1 TiB of
 * reservations: a search reserves 15-150 GiB, and the device's address space
 * for them is far smaller than the host's 47 bits.  scripts/vm_stress.py.)
 */
static inline void rt_place_retire(void *va, size_t bytes)
{
    rt_place_vm(hipMemAddressFree(va, bytes), "hipMemAddressFree");
}

/*
 * A mapping is only safe to launch a kernel on once the driver's page-table
 * update has landed, and hipMemMap + hipMemSetAccess return before that is
 * certain: about one process in eight that searched many pieces died of
 * "Memory access fault by GPU ... Reason: Unknown" (the GPU suite in one
 * process in round 5, bench.py in round 6 -- there on an address 0x45000
 * bytes into the SECOND slot of a search's scratch range, i.e. in the pair
 * test launched right after that slot had been mapped).  Allocating and
 * freeing a small buffer in between goes through the same driver queue
 * behind the update (it is also what invalidates stale translations, below):
 * every new mapping gets that before its first kernel.
 */
static inline void rt_place_settle_maps(void)
{
    void *t = NULL;
    if (hipMalloc(&t, (size_t)2 << 20) == hipSuccess)
        (void)hipFree(t);
    (void)hipGetLastError();
}

/* `flush`: the range was mapped -- whatever the device still holds of its
 * translations goes before the addresses can be handed out again (a plain
 * hipMalloc that follows may receive the same virtual addresses).
 * Callers that release several sets in a row flush once, after the last. */
static void rt_place_release(rt_place *p, bool flush = true)
{
    const bool was_mapped = p->base != NULL;
    if (p->base) {
        rt_place_vm(hipMemUnmap(p->base, p->bytes), "release: hipMemUnmap");
        rt_place_retire(p->base, p->bytes);
    }
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)p->handles;
    for (int k = 0; k < p->n; ++k)
        rt_place_vm(hipMemRelease(h[k]), "release: hipMemRelease");
    free(p->handles);
    memset(p, 0, sizeof *p);
    if (was_mapped && flush)
        rt_place_flush();
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

/*
 * A mapping the GPU has used stays in its address translation after
 * hipMemUnmap: kernels go on reading and writing the OLD physical memory
 * behind an address range that has been mapped again with other pieces,
 * while copies (the DMA engines) see the new one (round 5, ROCm 7.2;
 * scripts/labsrc/tlb_lab.hip, profiles/r05_probes/s19_*: the kernel's pattern
 * lands in the pieces mapped there BEFORE, every double of 4 GiB; re-reserving
 * the range is no cure across a few rounds, hipMemSetAccess is none; a plain
 * hipMalloc + hipFree in between is -- freeing a buffer makes the driver
 * invalidate the process's translations).  It is what made "the level
 * follows the address range" in this round's laboratories, and it would be
 * silent corruption in an engine that maps, measures and maps again.  So:
 * after EVERY hipMemMap sequence, before any kernel touches the range,
 * rt_place_flush(); and once the arrays have settled rt_place_coherent()
 * proves that kernels and copies see the same memory, or the context falls
 * back to plain allocations for good.
 */
static void rt_place_flush(void)
{
    void *t = NULL;
    if (hipMalloc(&t, (size_t)64 << 20) == hipSuccess)
        (void)hipFree(t);
    (void)hipGetLastError();
}

__global__ void rt_place_token_kernel(unsigned long long *base, size_t pages,
                                      size_t stride, unsigned long long salt)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pages)
        base[i * stride] = salt ^ (i * 0x9E3779B97F4A7C15ull);
}

/* one token per 2 MiB of the mapped range written by a kernel; the first,
 * the middle and the last of every piece read back by a copy (the path the
 * downloads take): true if every one arrives */
static bool rt_place_coherent(rt_ctx *c)
{
    const rt_place &P = c->place;
    if (!P.base)
        return true;
    const size_t page = (size_t)2 << 20, pages = P.bytes / page;
    const size_t per = P.piece / page;
    static unsigned long long round_ = 0;
    const unsigned long long salt = 0xC0FFEE1234567ull + 977 * ++round_;
    hipLaunchKernelGGL(rt_place_token_kernel,
                       dim3((unsigned)((pages + 255) / 256)), dim3(256), 0,
                       c->stream, (unsigned long long *)P.base, pages,
                       page / 8, salt);
    bool ok = hipStreamSynchronize(c->stream) == hipSuccess;
    for (int k = 0; ok && k < P.n; ++k) {
        const size_t at[3] = {(size_t)k * per, (size_t)k * per + per / 2,
                              (size_t)(k + 1) * per - 1};
        for (int q = 0; ok && q < 3; ++q) {
            unsigned long long got = 0;
            ok = hipMemcpy(&got, (const char *)P.base + at[q] * page, 8,
                           hipMemcpyDeviceToHost) == hipSuccess &&
                 got == (salt ^ (at[q] * 0x9E3779B97F4A7C15ull));
        }
    }
    (void)hipGetLastError();
    return ok;
}

/*
 * How many of the row streams a full trace writes at once fall into each
 * piece-sized slot of the range (round 6).  What the store pattern wants is
 * not an even mix of PIECES but of STREAMS: C2's 56 streams lie 18 / 19 / 11 /
 * 4 / 4 over its five slots, and the one piece of another class a set held
 * gave 6.85 TB/s in slot 0 or 1, 6.2-6.3 in slot 2 (where round-robin over the
 * classes put it: "A B C A B"), 5.5 in slot 3 or 4 -- the same five pieces
 * (scripts/c2_lab.py).  The layout is rt_reserve's plan: per block Y | U | I |
 * T planes of plan_L rows of plan_bs rays; a trace writes rows 1 .. L-1 of
 * Y, U and T (I where an element is tilted: not counted).
 */
static void rt_place_weights(const rt_ctx *c, size_t piece, int nslots,
                             float *w)
{
    for (int k = 0; k < nslots; ++k)
        w[k] = 0.f;
    const long long L = c->plan_L, bs = c->plan_bs;
    if (L < 2 || bs < 1)
        return;
    const long long plane = L * 3 * bs, bts = 10 * L * bs;
    auto add = [&](long long first) { /* a stream of bs doubles from `first` */
        const double b0 = 8. * (double)first, b1 = b0 + 8. * (double)bs;
        for (long long k = (long long)(b0 / (double)piece);
             k < nslots && (double)k * (double)piece < b1; ++k) {
            const double lo = (double)k * (double)piece;
            const double a = b0 > lo ? b0 : lo;
            const double z = b1 < lo + (double)piece ? b1 : lo + (double)piece;
            if (z > a)
                w[k] += (float)((z - a) / (b1 - b0));
        }
    };
    for (long long b = 0; b < (c->plan_nblk > 0 ? c->plan_nblk : 1); ++b)
        for (long long s = 1; s < L; ++s) {
            for (int q = 0; q < 3; ++q) {
                add(b * bts + (s * 3 + q) * bs);         /* Y */
                add(b * bts + plane + (s * 3 + q) * bs); /* U */
            }
            add(b * bts + 3 * plane + s * bs);           /* T */
        }
}

/*
 * The search is bounded in TIME, not only in pieces and bytes (VERDICT r5:
 * single hipMemCreate calls of 1.3-8.7 s were seen).  What an allocation must
 * have -- its `need` pieces -- it gets whatever that takes: the hipMalloc it
 * would fall back to waits for the same driver, and the stall that was seen
 * is the FIRST large hipMemCreate of a process (0.9-6 s on five of five boxes
 * of round 6, a millisecond for every piece after it), which says nothing
 * about the pieces that follow.  Everything that is CHOICE -- surplus pieces,
 * ballast hops, further sets of pieces -- ends RT_PLACE_BUDGET_MS after the
 * first set had the pieces it needs, and at once when a hipMemCreate of a
 * surplus piece, of ballast or of a further set takes RT_PLACE_STALL_MS.
 */
#define RT_PLACE_BUDGET_MS 250.
#define RT_PLACE_STALL_MS 200.
/* ... unless what there is so far is ONE class: such arrays write at 5.5
 * instead of 6.9 TB/s for as long as they live (round 6, pitch_lab3: a
 * single slow ballast block ended a search at [11, 0, 0]), so the search for
 * a second class goes on past the budget and past a stall, up to this many
 * milliseconds after the budget's clock started */
#define RT_PLACE_HARD_MS 3000.

static hipError_t rt_place_alloc(rt_ctx *c, void **out, size_t bytes)
{
    rt_place &P = c->place;
    memset(&P, 0, sizeof P);
    if (!c->opt_place || g_place_distrust || bytes < RT_PLACE_MIN_BYTES)
        return hipMalloc(out, bytes);
    const double t_start = rt_place_now_ms();
    /* the first set of an allocation starts the clock when it has what it
     * needs; rt_place_settle's further sets (all of them choice) run on what
     * is left of that budget */
    const bool choice = c->place_deadline_ms > 0.;
    double deadline = choice ? c->place_deadline_ms : 1e300;
    bool stalled = false, out_of_time = false;
    bool have_mix = false; /* `need` pieces could be picked, none of their
                              classes holding more than half */
    float slowest_create = 0.f;
    double t_ballast = 0.;

    /* pieces of 1 GiB (RT_MI355_PIECE_MIB: another size, for measurements);
     * arrays below 3 GiB: pieces of 512 MiB */
    size_t piece = (size_t)1 << 30;
    {
        const char *e = getenv("RT_MI355_PIECE_MIB");
        const long mib = e ? atol(e) : 0;
        if (mib >= 64 && mib <= 65536)
            piece = (size_t)mib << 20;
    }
    if (bytes < 3 * piece)
        piece >>= 1; /* 512 MiB: a power of two, i.e. ONE block of the
                        device's buddy allocator -- a 768 MiB piece is two
                        blocks that may lie in two classes, and its "one
                        piece" time is then already the fast one */
    const int need = (int)((bytes + piece - 1) / piece);
    size_t align = piece & (~piece + 1); /* largest power of two dividing it */
    /* pieces created and classified at most: large arrays may look further
     * for an even mix (C5, 97 pieces: as they came -- 22 / 45 / 30 -- the
     * last blocks of the batch lay in one class) */
    const int cap = need + (need > 48 ? need / 2 : 24);
    /* Pieces come in runs of one class (2-14 seen, 26+ on one box): once a
     * class is oversupplied the search HOPS -- a block of ballast is created
     * and held, unclassified, so that the next piece lies further on in the
     * device memory -- until another class turns up.  Ballast and surplus
     * pieces go back to the device before rt_place_alloc returns.  A hop is
     * four (after three hops in a row: eight) blocks of 1 GiB, not one block
     * of 4-8 GiB: creating ONE large block has taken 3.5 and 5.5 s where the
     * device memory was fragmented (round 5), small ones a millisecond. */
    const int max_ballast = 192;
    hipMemGenericAllocationHandle_t ballast[192];
    int nballast = 0, hops_in_a_row = 0, hops = 0;
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
    float *ratio = (float *)calloc(cap, sizeof(float)); /* (the log's) */
    void *scratch = NULL; /* every created piece at scratch + k * piece */
    int made = 0, mapped = 0, nclass = 0, rep[RT_PLACE_CLASSES];
    int count[RT_PLACE_CLASSES] = {0};
    float self_ms = 0.f, cross_ms = 0.f;
    hipError_t e = h && cls ? hipSuccess : hipErrorOutOfMemory;
    if (e == hipSuccess)
        e = hipMemAddressReserve(&scratch, (size_t)cap * piece, align, NULL, 0);
    /* the range may be one an earlier allocation of the process has used:
     * nothing of that may be left in the device when the pair tests write
     * through it (every slot below is then mapped exactly once) */
    rt_place_flush();
    bool enough = false;
    while (e == hipSuccess && made < cap && !enough) {
        const int k = made;
        if (made >= need && extra + piece > budget)
            break; /* the surplus has reached its share of the free memory */
        if (made >= need &&
            (stalled || (out_of_time = rt_place_now_ms() > deadline)) &&
            (have_mix || choice ||
             rt_place_now_ms() > deadline + RT_PLACE_HARD_MS))
            break; /* no time left to choose: what exists must do */
        const double t_create = rt_place_now_ms();
        if (hipMemCreate(&h[k], piece, &prop, 0) != hipSuccess) {
            (void)hipGetLastError();
            break; /* the device is full: what exists must do */
        }
        {
            const float took = (float)(rt_place_now_ms() - t_create);
            slowest_create = took > slowest_create ? took : slowest_create;
            /* (a piece the arrays need is no reason to stop choosing) */
            stalled = stalled ||
                      (took > RT_PLACE_STALL_MS && (choice || made >= need));
            /* (while what there is is ONE class the clock stops for a stall:
             * the hard limit is for the search's own work -- a 4 s create in
             * a 20 GiB stretch of one class had used it up, round 6) */
            if (took > RT_PLACE_STALL_MS && !have_mix && deadline < 1e299) {
                deadline += took;
                c->place_deadline_ms = deadline;
            }
        }
        ++made;
        if (!choice && made == need)
            c->place_deadline_ms = deadline =
                rt_place_now_ms() + c->opt_place_budget_ms;
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
        rt_place_settle_maps(); /* before the first kernel on this slot */
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
                if (ratio && ms / self_ms > ratio[k])
                    ratio[k] = ms / self_ms; /* the slowest pairing seen */
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
        /* a class is oversupplied: more than half of what is needed while
         * only two classes have shown, more than 40 % once it is about the
         * third */
        const int over = nclass >= 2 && made >= need ? (2 * need + 4) / 5
                                                     : (need + 1) / 2;
        /* two classes and eight pieces beyond the need without a third: a
         * first set of 4 GiB and more reaches FURTHER for it -- the classes
         * are thirds of the device memory handed out in runs of up to 64 GiB
         * (scripts/class_map.py: AAAAAAAABBBBBBBBBBBBBCCCCC... on five
         * boxes), and on a box whose first 60 GiB held two of them the
         * headline ran 4 % behind (round 6, box a: [5, 5, 0] at 6.64 TB/s,
         * 1.026 ms) -- with a hop of 8 GiB before every further piece, while
         * the time and the memory budget last */
        const bool seek_third = !choice && nclass == 2 && made >= need + 8 &&
                                (size_t)need * piece >= ((size_t)4 << 30);
        if ((count[cls[k]] > over || seek_third) &&
            made >= (need + 1) / 2 + 1 && hops < 24 &&
            ((!stalled && rt_place_now_ms() < deadline) ||
             (!have_mix && !choice &&
              rt_place_now_ms() < deadline + RT_PLACE_HARD_MS))) {
            const int blocks = hops_in_a_row < 3 && !seek_third ? 4 : 8;
            const size_t one = (size_t)1 << 30;
            int hopped = 0;
            for (int b = 0; b < blocks && nballast < max_ballast &&
                            extra + one <= budget &&
                            (t_ballast < 40. || !have_mix || seek_third) &&
                            (!stalled || b == 0); ++b) {
                const double tb = rt_place_now_ms();
                if (hipMemCreate(&ballast[nballast], one, &prop, 0) !=
                    hipSuccess) {
                    (void)hipGetLastError(); /* the device is full */
                    break;
                }
                ++nballast;
                ++hopped;
                extra += one;
                const double took = rt_place_now_ms() - tb;
                t_ballast += took;
                slowest_create =
                    (float)took > slowest_create ? (float)took : slowest_create;
                stalled = stalled || took > RT_PLACE_STALL_MS;
                if (took > RT_PLACE_STALL_MS && !have_mix && deadline < 1e299) {
                    deadline += took;
                    c->place_deadline_ms = deadline;
                }
            }
            if (hopped) { /* (a hop that created nothing is not one) */
                ++hops;
                ++hops_in_a_row;
            } else {
                hops_in_a_row = 0;
            }
        } else {
            hops_in_a_row = 0;
        }
        /* enough when `need` pieces can be picked with no class holding more
         * than half of them -- and THREE classes are among them, or eight
         * pieces beyond the need have not turned up a third (sets of two
         * classes: 6.67-6.77 TB/s in five of five contexts, sets of three:
         * 6.91-7.02 in nine of ten, profiles/r05_final/boxstat/) */
        if (made >= need && nclass >= 2) {
            /* (with three classes: none above 40 % of the need, as long as
             * the search may go on) */
            const int most = nclass >= 3 && made < cap - 1 ? (2 * need + 4) / 5
                                                          : (need + 1) / 2;
            int can = 0, can2 = 0;
            for (int q = 0; q < nclass; ++q) {
                can += count[q] < most ? count[q] : most;
                can2 += count[q] < (need + 1) / 2 ? count[q] : (need + 1) / 2;
            }
            have_mix = can2 >= need;
            /* (two classes: eight pieces beyond the need end the search of a
             * small or a further set; a large first set goes on by hops of
             * 8 GiB until the third shows or hops, memory or time run out) */
            const bool reach_over =
                !seek_third || hops >= 24 || stalled ||
                extra + ((size_t)9 << 30) > budget ||
                rt_place_now_ms() > deadline;
            enough = can >= need &&
                     (nclass >= 3 || (made >= need + 8 && reach_over));
        }
    }
    /* RT_MI355_PLACE_LOG=1: the classes in the order the pieces were created
     * (walked without hops, a fresh process shows runs of 4 / 8 / 15 ...
     * pieces of 1 GiB: AAAAAAAABBBBCAAAABBBBBBBBAAAAAAAACCCCBBBBBBBBCCCC...,
     * profiles/r05_final/placement_class_map_60_pieces.txt) */
    if (getenv("RT_MI355_PLACE_LOG")) {
        char line[512];
        int at = 0;
        for (int k = 0; k < made && at < 500; ++k)
            line[at++] = (char)('A' + cls[k]);
        line[at] = 0;
        fprintf(stderr, "[rt_place] need %d made %d classes %d hops %d: %s\n",
                need, made, nclass, hops, line);
        /* the slowest pairing of every piece against the classes it was
         * tested with, in units of the one-piece time (above 0.91: same
         * class) */
        if (ratio && made <= 40) {
            at = 0;
            for (int k = 0; k < made && at < 500; ++k)
                at += snprintf(line + at, sizeof line - at, " %.2f", ratio[k]);
            fprintf(stderr, "[rt_place]   pair / self:%s\n", line);
        }
    }
    free(ratio);
    const double t_found = rt_place_now_ms(), t_created = t_ballast;
    for (int b = 0; b < nballast; ++b)
        rt_place_vm(hipMemRelease(ballast[b]), "search: ballast hipMemRelease");
    t_ballast += rt_place_now_ms() - t_found;
    if (e != hipSuccess || made < need) {
        /* not this way: give everything back, allocate plainly */
        (void)hipGetLastError();
        for (int k = 0; k < mapped; ++k)
            rt_place_vm(hipMemUnmap((char *)scratch + (size_t)k * piece, piece),
                        "search gave up: hipMemUnmap");
        for (int k = 0; k < made; ++k)
            rt_place_vm(hipMemRelease(h[k]), "search gave up: hipMemRelease");
        if (scratch)
            rt_place_retire(scratch, (size_t)cap * piece);
        free(h);
        free(cls);
        if (mapped) /* the plain buffer may get these very addresses */
            rt_place_flush();
        return hipMalloc(out, bytes);
    }
    /* pick `need` pieces so that the STREAMS are mixed evenly over the
     * classes: slots in the order of their stream counts, each to the class
     * that has been given the least so far and still has a piece (where the
     * plan is unknown or flat this is the round-robin of rounds 4-5) */
    /* (the early ways out above leave `ratio` to this free) */
    int *pick = (int *)calloc(need, sizeof(int));
    hipMemGenericAllocationHandle_t *kept =
        (hipMemGenericAllocationHandle_t *)calloc(need, sizeof *kept);
    float *wslot = (float *)calloc(need, sizeof(float));
    int *by_weight = (int *)calloc(need, sizeof(int));
    unsigned char *slot_q = (unsigned char *)calloc(need, 1);
    unsigned char slot_class[64] = {0};
    int next[RT_PLACE_CLASSES] = {0}, taken = 0;
    int used[RT_PLACE_CLASSES] = {0};
    if (pick && kept && wslot && by_weight && slot_q) {
        rt_place_weights(c, piece, need, wslot);
        for (int k = 0; k < need; ++k) { /* insertion sort, heaviest first;
                                            equal weights keep their order */
            int j = k;
            while (j > 0 && wslot[by_weight[j - 1]] < wslot[k]) {
                by_weight[j] = by_weight[j - 1];
                --j;
            }
            by_weight[j] = k;
        }
        double given[RT_PLACE_CLASSES] = {0.};
        int left[RT_PLACE_CLASSES];
        for (int q = 0; q < RT_PLACE_CLASSES; ++q)
            left[q] = q < nclass ? count[q] : 0;
        int rr = 0; /* ties go round the classes */
        for (int i = 0; i < need; ++i) {
            const int slot = by_weight[i];
            int best = -1;
            for (int d = 0; d < nclass; ++d) {
                const int q = (rr + d) % nclass;
                if (left[q] > 0 && (best < 0 || given[q] < given[best] - 1e-9))
                    best = q;
            }
            if (best < 0)
                break;
            slot_q[slot] = (unsigned char)best;
            given[best] += wslot[slot] > 0.f ? wslot[slot] : 1e-3;
            --left[best];
            rr = (best + 1) % nclass;
            ++taken;
        }
        if (taken == need)
            for (int slot = 0; slot < need; ++slot) {
                const int q = slot_q[slot];
                int k = next[q];
                while (k < made && cls[k] != q)
                    ++k;
                pick[slot] = k; /* (k < made: left[] counted them) */
                next[q] = k + 1;
                ++used[q];
                if (slot < 64)
                    slot_class[slot] = (unsigned char)q;
            }
    }
    free(wslot);
    free(by_weight);
    free(slot_q);
    for (int k = 0; k < made; ++k)
        rt_place_vm(hipMemUnmap((char *)scratch + (size_t)k * piece, piece),
                    "search: scratch hipMemUnmap");
    /* the scratch range goes back first: the final range then begins where
     * the pair tests ran */
    rt_place_retire(scratch, (size_t)cap * piece);
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
    if (e == hipSuccess)
        rt_place_flush();
    for (int k = 0; k < made; ++k) /* the surplus */
        if (h[k])
            rt_place_vm(hipMemRelease(h[k]), "search: surplus hipMemRelease");
    free(h);
    free(cls);
    free(pick);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        for (int k = 0; k < nm && kept; ++k) {
            if (kept[k]) { /* mapped */
                rt_place_vm(hipMemUnmap((char *)base + (size_t)k * piece,
                                        piece),
                            "mapping failed: hipMemUnmap");
                rt_place_vm(hipMemRelease(kept[k]),
                            "mapping failed: hipMemRelease");
            }
        }
        if (base)
            rt_place_retire(base, (size_t)need * piece);
        free(kept);
        rt_place_flush(); /* the scratch slots were mapped and are gone */
        return hipMalloc(out, bytes);
    }
    P.base = base;
    P.bytes = (size_t)need * piece;
    P.piece = piece;
    P.n = need;
    P.handles = kept;
    for (int k = 0; k < need && k < 64; ++k)
        P.slot_cls[k] = slot_class[k];
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
    P.ballast = hops;
    P.slowest_create_ms = slowest_create;
    P.cut_short = stalled ? 2 : out_of_time ? 1 : 0;
    P.class_mix = P.mixed;
    P.fast = P.mixed; /* (until the pattern itself has been measured) */
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
 * The batch's own pattern over the arrays where they are: what decides
 * between four and two workgroups per CU (rt_resident_lds) is this
 * measurement, not the classes.  (Mapping the same pieces behind another
 * address range and measuring again was built and removed in round 5: the
 * ranges of one set of pieces lie within 2 % of each other, and what had
 * looked like more in the laboratories were stale translations,
 * rt_place_flush.)
 */
static void rt_place_tune(rt_ctx *c, int L, long long ld)
{
    rt_place &P = c->place;
    if (!P.base || L < 2 || 56. * (L - 1) * (double)ld < 5e8)
        return;
    const double t_start = rt_place_now_ms();
    P.store_gbps = rt_place_measure(c, L, ld);
    if (P.store_gbps > 0.f) /* whatever the classes said */
        P.fast = P.store_gbps >= RT_PLACE_FAST_GBPS;
    P.tune_ms = (float)(rt_place_now_ms() - t_start);
}

/*
 * The pieces of the current set in another order along a range of their own
 * (perm[k] = index of the piece that goes to slot k).  The arrays hold
 * nothing yet (rt_reserve, fresh layout); c->d_buf follows.
 */
static hipError_t rt_place_reorder(rt_ctx *c, const int *perm)
{
    rt_place &P = c->place;
    hipMemGenericAllocationHandle_t *h =
        (hipMemGenericAllocationHandle_t *)P.handles;
    hipMemGenericAllocationHandle_t *nh =
        (hipMemGenericAllocationHandle_t *)calloc(P.n, sizeof *nh);
    if (!nh)
        return hipErrorOutOfMemory;
    hipMemAccessDesc acc = {};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = c->device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    /* a NEW range for the new order (an address is mapped once in its life) */
    void *base = NULL;
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess)
        e = hipMemAddressReserve(&base, P.bytes, P.piece & (~P.piece + 1), NULL,
                                 0);
    if (e == hipSuccess) {
        e = hipMemUnmap(P.base, P.bytes);
        if (e != hipSuccess)
            rt_place_retire(base, P.bytes);
    }
    if (e == hipSuccess) {
        rt_place_retire(P.base, P.bytes);
        for (int k = 0; k < P.n && e == hipSuccess; ++k) {
            nh[k] = h[perm[k]];
            e = hipMemMap((char *)base + (size_t)k * P.piece, P.piece, 0,
                          nh[k], 0);
        }
        if (e == hipSuccess)
            e = hipMemSetAccess(base, P.bytes, &acc, 1);
        memcpy(h, nh, P.n * sizeof *nh);
        {
            unsigned char lab[64];
            memcpy(lab, P.slot_cls, sizeof lab);
            for (int k = 0; k < P.n && k < 64; ++k)
                P.slot_cls[k] = perm[k] < 64 ? lab[perm[k]] : 0;
        }
        P.base = base;
        c->d_buf = (double *)base;
        rt_place_flush();
        if (e != hipSuccess) { /* no arrays any more: everything goes back */
            (void)hipGetLastError();
            rt_place_release(&c->place);
            c->d_buf = NULL;
        }
    }
    free(nh);
    return e;
}

/*
 * The SAME pieces in another order along the range (round 6).  Which piece
 * lies under which part of the arrays decides as much as which pieces they
 * are: five 512 MiB pieces of C2 ran the pattern at 6166, 6080, 6088, 6229,
 * 5495, 6820 and 5541 GB/s in seven orders (scripts/c2_lab.py with
 * RT_MI355_PLACE_PERM; three contexts of eight like that, the others between
 * 6717 and 7067), ten 1 GiB pieces of the headline batch between 6483 and
 * 7026 with the class-interleaved order near the top.  An order costs a
 * remap into a range of its own (0.7 ms; an address is mapped once) and one
 * measurement of the pattern (4.5 ms) -- no memory, no search -- so orders are
 * tried before another set of pieces is: up to `tries`, seeded permutations,
 * while the pattern is below the mark and the allocation's time lasts; the
 * best order stays.  Returns the orders tried.
 */
static int rt_place_orders(rt_ctx *c, int L, long long ld, int tries)
{
    const int n = c->place.n;
    if (tries < 1 || n < 2 || n > 64 || !c->place.base ||
        !(c->place.store_gbps > 0.f))
        return 0;
    int best[64], cur[64], trial[64], now[64];
    for (int k = 0; k < n; ++k)
        best[k] = cur[k] = k;
    float best_gbps = c->place.store_gbps;
    const bool log = getenv("RT_MI355_PLACE_LOG") != NULL;
    auto labels = [&](char *dst) {
        for (int k = 0; k < n && k < 63; ++k)
            dst[k] = (char)('A' + c->place.slot_cls[k]);
        dst[n < 63 ? n : 63] = 0;
    };
    char lab[64];
    if (log && n <= 12) {
        /* every pair of the set's pieces: launch time over the one-piece
         * time (rows = slots; above 0.91: "same class") */
        const long long np =
            (long long)(c->place.piece / sizeof(double) / RT_PLACE_ROWS) /
            256 * 256;
        for (int i = 0; i < n; ++i) {
            fprintf(stderr, "[rt_place]   pair matrix %d:", i);
            float self = 0.f;
            double *pi = (double *)((char *)c->place.base +
                                    (size_t)i * c->place.piece);
            (void)rt_place_time(c, pi, pi, np, &self);
            for (int j = 0; j < n; ++j) {
                float ms = 0.f;
                double *pj = (double *)((char *)c->place.base +
                                        (size_t)j * c->place.piece);
                (void)rt_place_time(c, pi, pj, np, &ms);
                fprintf(stderr, " %.2f", self > 0.f ? ms / self : 0.f);
            }
            fprintf(stderr, "\n");
        }
    }
    if (log) {
        labels(lab);
        fprintf(stderr, "[rt_place]   orders: %s %.0f", lab, best_gbps);
    }
    unsigned rs = 12345u + 977u * (unsigned)n;
    int tried = 0;
    for (int t = 0; t < tries && best_gbps < c->opt_place_good &&
                    rt_place_now_ms() < c->place_deadline_ms; ++t) {
        for (int k = 0; k < n; ++k) /* relative to the current mapping */
            trial[k] = k;
        for (int k = n - 1; k > 0; --k) {
            rs = rs * 1664525u + 1013904223u;
            const int j = (int)((rs >> 8) % (unsigned)(k + 1));
            const int x = trial[k];
            trial[k] = trial[j];
            trial[j] = x;
        }
        if (rt_place_reorder(c, trial) != hipSuccess)
            return -1; /* the context has lost its arrays */
        for (int k = 0; k < n; ++k)
            now[k] = cur[trial[k]];
        memcpy(cur, now, sizeof(int) * n);
        rt_place_tune(c, L, ld);
        ++tried;
        if (log) {
            labels(lab);
            fprintf(stderr, " | %s %.0f", lab, c->place.store_gbps);
        }
        if (c->place.store_gbps > best_gbps) {
            best_gbps = c->place.store_gbps;
            memcpy(best, cur, sizeof(int) * n);
        }
    }
    if (memcmp(best, cur, sizeof(int) * n)) { /* back to the best order */
        int inv[64];
        for (int k = 0; k < n; ++k)
            inv[cur[k]] = k;
        for (int k = 0; k < n; ++k)
            trial[k] = inv[best[k]];
        if (rt_place_reorder(c, trial) != hipSuccess)
            return -1;
        c->place.store_gbps = best_gbps; /* (measured when it was tried) */
        c->place.fast = best_gbps >= RT_PLACE_FAST_GBPS;
    }
    if (log)
        fprintf(stderr, " -> %.0f\n", c->place.store_gbps);
    return tried;
}

/*
 * rt_place_tune, and while the arrays stay below RT_PLACE_GOOD_GBPS ANOTHER
 * set of pieces: the current one is held (so that the new pieces come from
 * elsewhere in the device memory), classified, mapped and measured like the
 * first, and the better set stays.  At most three sets -- five for arrays
 * below 4 GiB (a set costs them ~10 ms and is bad in one of three cases), and
 * where the best of three is still 2.5 % below the mark --, arrays up to
 * 16 GiB (C2: 0.2414 ms behind pieces whose four ranges all ran the pattern
 * at 5.45-5.5 TB/s, 0.207-0.218 behind others).  ctx->d_buf follows.
 */
#define RT_PLACE_PICKS 8 /* (rt_placement reports the first five) */
static hipError_t rt_place_alloc(rt_ctx *c, void **out, size_t bytes);

static void rt_place_settle(rt_ctx *c, int L, long long ld, size_t bytes)
{
    if (!c->place.base || L < 2 || 56. * (L - 1) * (double)ld < 5e8)
        return; /* (a pattern too short to tell anything: what an earlier
                   layout found out about these arrays stays) */
    const double t_settle = rt_place_now_ms();
    rt_place_tune(c, L, ld);
    /* orders of the first set before any further set (arrays up to 16 GiB) */
    const int order_tries =
        getenv("RT_MI355_PLACE_PERM") ? atoi(getenv("RT_MI355_PLACE_PERM"))
        : c->opt_place_orders >= 0    ? c->opt_place_orders
        : c->place.bytes < ((size_t)4 << 30) ? 6 : 3;
    int orders = 0;
    if (c->place.bytes <= ((size_t)16 << 30)) {
        const int k = rt_place_orders(c, L, ld, order_tries);
        if (k < 0)
            return; /* (c->d_buf is NULL: rt_reserve reports it) */
        orders += k;
    }
    int picks = 1, nlost = 0;
    float seen[RT_PLACE_PICKS] = {c->place.store_gbps};
    rt_place lost[RT_PLACE_PICKS];
    float slowest = c->place.slowest_create_ms;
    int cut = c->place.cut_short;
    /* at most three sets for arrays of 4 GiB and more (a set costs them
     * 10-20 ms) -- five where the best of three is still 2.5 % below the
     * mark (three sets of two classes in a row: 6.59-6.63 TB/s, C3' 0.77
     * instead of 0.82, profiles/r05_final/boxstat/bench_box_f.json) */
    /* Round 6: sets drawn one after the other come out alike (all five of a
     * context at 5.5-6.4 TB/s on two boxes of eight, 6.9-7.0 at the first
     * draw on the others; sets of 256 MiB pieces, which the pair test cannot
     * tell apart at all, show the same two levels: scripts/c2_lab.py) --
     * what decides lies beyond the classes, so the draws are made cheaper
     * and further apart: arrays below 4 GiB (a set costs them 5-10 ms) get
     * up to eight sets inside the same time budget, and two GiB of ballast
     * are put between one set and the next. */
    const bool small = c->place.bytes < ((size_t)4 << 30);
    hipMemGenericAllocationHandle_t gap[8 * RT_PLACE_PICKS];
    int ngap = 0;
    while (picks < (small ? RT_PLACE_PICKS : 5) && c->d_buf &&
           c->place.base && c->place.store_gbps > 0.f &&
           c->place.store_gbps < c->opt_place_good &&
           (picks < 3 || small ||
            c->place.store_gbps < .975 * c->opt_place_good) &&
           c->place.bytes <= ((size_t)16 << 30)) {
        /* another set is choice, not need: none once the allocation's time is
         * up or a hipMemCreate has stalled */
        if (cut == 2 || rt_place_now_ms() > c->place_deadline_ms) {
            /* ... unless the arrays behave like ONE class (a fifth slower for
             * as long as they live): those get another search with a budget
             * of its own, while the hard limit lasts */
            if (!(c->place.store_gbps < RT_PLACE_FAST_GBPS) ||
                rt_place_now_ms() > t_settle + RT_PLACE_HARD_MS) {
                cut = cut ? cut : 1;
                break;
            }
            c->place_deadline_ms = rt_place_now_ms() + c->opt_place_budget_ms;
        }
        const rt_place held = c->place; /* pieces and range stay alive */
        double *const held_buf = c->d_buf;
        if (small) { /* the next set from further on in the device memory */
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = c->device;
            /* (eight GiB where the set behaves like ONE class: such
             * stretches are 4 ... 64 GiB long) */
            const int blocks =
                c->place.store_gbps < RT_PLACE_FAST_GBPS ? 8 : 2;
            for (int b = 0; b < blocks && ngap < 8 * RT_PLACE_PICKS; ++b) {
                if (hipMemCreate(&gap[ngap], (size_t)1 << 30, &prop, 0) !=
                    hipSuccess) {
                    (void)hipGetLastError();
                    break;
                }
                ++ngap;
            }
        }
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
        {   /* (a further set gets two orders of its own) */
            const int k = rt_place_orders(c, L, ld,
                                          order_tries < 2 ? order_tries : 2);
            if (k < 0) { /* the new set is gone: the held one stays */
                c->place = held;
                c->d_buf = held_buf;
                break;
            }
            orders += k;
        }
        seen[picks++] = c->place.store_gbps;
        slowest = c->place.slowest_create_ms > slowest
                      ? c->place.slowest_create_ms : slowest;
        cut = c->place.cut_short > cut ? c->place.cut_short : cut;
        /* the loser stays mapped until the search is over: given back now,
         * its pieces would be the first the next search is handed -- five
         * sets out of the same stretch of the device memory (two classes
         * five times in a row on one box: 6.51-6.69 TB/s) */
        if (!c->d_buf || c->place.store_gbps <= held.store_gbps) {
            if (c->place.base || c->place.handles)
                lost[nlost++] = c->place; /* no better: back to the held */
            c->place = held;
            c->d_buf = held_buf;
        } else {
            lost[nlost++] = held;
        }
    }
    for (int k = 0; k < nlost; ++k)
        rt_place_release(&lost[k], false); /* (one flush, below) */
    for (int b = 0; b < ngap; ++b)
        rt_place_vm(hipMemRelease(gap[b]), "settle: gap hipMemRelease");
    if (getenv("RT_MI355_PLACE_LOG")) {
        fprintf(stderr, "[rt_place]   store pattern per set:");
        for (int k = 0; k < picks; ++k)
            fprintf(stderr, " %.0f", seen[k]);
        fprintf(stderr, " GB/s, kept %.0f, [%d %d %d] of %d x %zu MiB\n",
                c->place.store_gbps, c->place.count[0], c->place.count[1],
                c->place.count[2], c->place.n, c->place.piece >> 20);
    }
    c->place.picks = picks;
    c->place.orders = orders;
    c->place.slowest_create_ms = slowest;
    c->place.cut_short = cut;
    c->place.settled = 1;
    /* (five slots in rt_placement: the fifth holds the best of the fifth
     * and later sets) */
    for (int k = 0; k < 5; ++k)
        c->place.pick_gbps[k] = k < picks ? seen[k] : 0.f;
    for (int k = 5; k < picks; ++k)
        if (seen[k] > c->place.pick_gbps[4])
            c->place.pick_gbps[4] = seen[k];
    /* sets that lost have been unmapped: whatever the device still holds of
     * their translations goes before anything else is launched */
    rt_place_flush();
}

#endif /* RT_PLACE_H */
