/*
 * rt_engine.hip -- core of librt_mi355.so (C ABI: include/rt_mi355.h), the
 * gfx950 engine of the sequential geometric ray trace: contexts, surface
 * tables, seeding / generation of launch rays, the trace launch, the
 * bookkeeping of rows that are served instead of stored, downloads.
 * Consumers (rms, opd, aiming ...) are in rt_consumers.hip, the RCCL gather in
 * rt_comm.hip, kernels in rt_trace_kernels.h, per-ray arithmetic in rt_math.h.
 *
 * Kernel design (MI355X first):
 *  - one lane owns one ray (8-byte global accesses, 512 B per wave
 *    instruction; 2 and 4 rays per lane were measured slower) and
 *    keeps its state (y, u) in VGPRs across the whole surface loop: the fused
 *    march reads 48 B per ray once and writes 56-80 B per ray-surface op,
 *    nothing is ever re-read;
 *  - results are SoA [surface][component][ray] so every store instruction of
 *    a wave covers one contiguous, 512 B aligned segment; rows that are bit
 *    for bit another row (i[j] = u[j-1] without tilts, u[j] = i[j] where
 *    nothing bends or clips) are served, not stored;
 *  - the surface table is wave-uniform: it is read with scalar loads
 *    (s_load_dwordx*) through the scalar cache into SGPRs, costing no VGPRs
 *    and no LDS traffic; all per-surface branches are scalar branches; it is
 *    double buffered on the device so a changed table never drains the stream;
 *  - the even-asphere Newton solve is the only divergent loop; its trip count
 *    is decided per wavefront with a 64-bit ballot (rt_math.h); by default it
 *    runs on FMA / rcp / rsq with one reciprocal per iterate (1e-8 contract),
 *    which takes it off the FP64-issue wall; the bit-for-bit restatement of
 *    scipy's iteration is an option;
 *  - a wavefront whose rays are all dead stores NaN rows without evaluating
 *    the element; an opt-in kernel compacts the survivors of a workgroup into
 *    fewer wavefronts (ballots + LDS) for traces that keep few rows;
 *  - FP64 VALU only: the path is elementwise, there is no contraction to put
 *    on MFMA.  Bound: HBM write bandwidth (56-80 B / ray-surface op).
 *
 * No CPU fallback lives here: every entry point either runs on the GPU or
 * returns an error.
 */
#include <thread>

#include "rt_ctx.h"
#include "rt_trace_kernels.h"
#include "rt_place.h"
#include "rt_copy_pool.h"

static char g_err[512] = "";
int g_place_distrust = 0;

int rt_fail(rt_ctx *ctx, int code, const char *fmt, ...)
{
    char *dst = ctx ? ctx->err : g_err;
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(dst, 512, fmt, ap);
    va_end(ap);
    if (ctx)
        snprintf(g_err, sizeof g_err, "%s", ctx->err);
    return code;
}

/* give row `surf` of U or I its own copy of the data it is served from (a
 * kernel is about to read it at its natural address, or the row it is served
 * from is about to be overwritten) */
int rt_detach(rt_ctx *c, int which, int surf)
{
    unsigned char *alias = which == RT_U ? c->u_alias : c->i_alias;
    if ((which != RT_U && which != RT_I) || !alias[surf])
        return RT_OK;
    const double *src = rt_row(c, which, surf);
    alias[surf] = 0;
    double *dst = rt_row(c, which, surf);
    for (int b = 0; b < c->nblk; ++b) /* the three components, block by block */
        RT_HIP(c, hipMemcpyAsync(dst + (size_t)b * c->bts,
                                 src + (size_t)b * c->bts,
                                 (size_t)3 * c->bs * sizeof(double),
                                 hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

/* the notes on row 0's tiles as the kernels take them (rt_trace_kernels.h),
 * starting at tile `t0` */
static inline size_t rt_tiles_bytes(size_t tiles)
{
    return ((tiles + 1) / 2 * 2) * sizeof(unsigned) +
           6 * tiles * sizeof(double);
}

static inline rt_tiles rt_tiles_of(const rt_ctx *c, int64_t t0, bool on)
{
    rt_tiles t = {NULL, NULL, 0};
    if (on && c->d_uni) {
        const size_t cap = c->uni_cap;
        t.note = c->d_uni + t0;
        t.first = (double *)(c->d_uni + (cap + 1) / 2 * 2) + t0;
        t.stride = (int64_t)cap;
    }
    return t;
}

/* row 0 of a generated batch, if no trace has built it yet */
int rt_gen_flush(rt_ctx *c)
{
    if (!c || !c->gen_pending)
        return RT_OK;
    c->gen_pending = 0;
    c->uni_valid = 0; /* row 0 is rewritten */
    RT_HIP(c, hipSetDevice(c->device));
    hipLaunchKernelGGL(rt_generate_kernel,
                       dim3((unsigned)((c->ld + 255) / 256)), dim3(256), 0,
                       c->stream, (const rt_field *)c->d_gen,
                       (const double *)((char *)c->d_gen + c->gen_fpad),
                       c->gen_np, c->gen_n, c->gen_s0, rt_layout(c), c->ld,
                       !c->opt_alias);
    RT_HIP(c, hipGetLastError());
    return RT_OK;
}

/*
 * Unused dynamic LDS per workgroup of the trace kernels = a cap on the
 * workgroups resident per CU (160 KB / bytes; the registers allow seven).
 * A trace that stores its rows is bound by the memory side, and with TWO
 * workgroups per CU the chip writes a more compact window of every row stream
 * at any moment: measured never slower and 0.2-6 % faster depending on the
 * kind of trace and on the box (host-seeded C3 +0.2 % on slow boxes, +5 % on
 * a fast one; unclipped +3 %; device-generated +6 %; C2 +3 %).  Traces that
 * keep few rows live on FP64 issue and want every wavefront (image row only:
 * 0.60 -> 0.81 ms with two workgroups); the Newton solves of aspheric
 * elements sit in between: four workgroups per CU on the default arithmetic
 * (+1.4 %), no cap on the exact one.  profiles/r03_probes/README.md.
 */
static size_t rt_resident_lds(const rt_ctx *c, int start, int stop)
{
    if (c->opt_resident >= 0)
        return (size_t)c->opt_resident;
    int stored = 0, stored_i = 0, newton = 0;
    for (int s = start; s < stop; ++s) {
        const unsigned f = c->h_stage[s].flags;
        stored += !(f & RT_F_NOSTORE);
        stored_i += (f & (RT_F_NOSTORE | RT_F_STORE_I)) == RT_F_STORE_I;
        newton += (f & RT_F_ASPH) != 0;
    }
    if (2 * stored < stop - start)
        return 0;
    if (newton) /* default arithmetic: six per CU in mixed memory (0.88-0.94
                   ms on two boxes; five: 0.88-0.95, four: 0.90-0.98, no cap:
                   0.88-0.94), four where the memory is of one class */
        return c->opt_fast ? (c->place.fast ? 24576 : 32768) : 0;
    /* store bound: four workgroups per CU where the arrays lie in a mix of
     * memory classes (rt_place.h: 1.08 ms against 1.23 with two; three: 1.09,
     * five: 1.13), two where they do not -- two per CU is the setting that
     * does not care where it writes (1.22-1.25 ms in any allocation; four:
     * 1.35 in a bad one) -- and where most elements store their i rows too
     * (tilted systems: ten streams per element, 1.49 against 1.52) */
    if (2 * stored_i > stored)
        return 65536;
    if (c->place.fast) {
        /* ... unless the launch rows are read ray by ray (hardly a component
         * uniform across a tile: a caller's own bundle): the kernel without
         * the Newton solves leaves room for eleven wavefronts per SIMD, and
         * the loads among the saturated stores want them -- C3' 1.080-1.087
         * ms with eight and more workgroups per CU against 1.093-1.101 with
         * four, the collimated headline 0.976-0.979 against 0.965-0.980
         * (scripts/c3p_lab.py, round 6) */
        if (start == 1 && c->uni_valid && c->uni_share >= 0.f &&
            c->uni_share < .4f && !c->table_asph)
            return 0;
        return 32768;
    }
    /* plain allocations: two per CU (the setting that does not care where
     * it writes), except small batches, short kernels whose launch ramp
     * wants every wavefront (3*10^5 rays: 0.049 ms uncapped, 0.058 with two
     * per CU) -- profiles/r04_final/nsweep.jsonl */
    if ((size_t)c->cap_doubles * sizeof(double) < RT_PLACE_MIN_BYTES &&
        c->n < ((int64_t)1 << 19))
        return 0;
    return 65536;
}

/* the result arrays: class-mixed pieces (rt_place.h); the laboratory build
 * can ask for other kinds of allocation instead */
static hipError_t rt_buf_alloc(rt_ctx *c, void **out, size_t bytes)
{
    return rt_place_alloc(c, out, bytes);
}

static hipError_t rt_buf_free(rt_ctx *c, void *p)
{
    return rt_place_free(c, p);
}

static bool rt_in_range(double x)
{
    const double a = fabs(x);
    return a >= RT_RANGE_TINY && a <= 0x1p99; /* strictly inside the guard */
}

/* whatever replaces the launch rays, a row or the surface table ends a step
 * that was being traced in pieces (rt_trace_chunk): the pieces traced so far
 * belong to the old batch / table, and pieces of the new one must not count
 * as the rest of the old step */
static inline void rt_pieces_reset(rt_ctx *c)
{
    c->opd_n = 0; /* path differences kept on the device (rt_opd_device) are
                     those of the rows as they were: every caller of this is
                     about to replace rays, a row or the table */
    c->pieces_seen = c->pieces_total = 0;
    memset(c->pieces_mask, 0, sizeof c->pieces_mask);
}

/* the compacting variant pays (one barrier per element) only where dead rays
 * are wasted FP64 issue, i.e. where rows are traced but not stored */
static bool rt_use_compact(const rt_ctx *c, int start, int stop)
{
    if (!c->opt_compact)
        return false;
    if (c->ngroups > 1 && (c->n / c->ngroups) % RT_CB)
        return false; /* a 256-ray tile would straddle two tables */
    if (c->opt_compact == 2)
        return true;
    for (int s = start; s < stop; ++s)
        if (c->h_stage[s].flags & RT_F_NOSTORE)
            return true;
    return false;
}

static inline void rt_event_free(hipEvent_t e)
{
    if (e)
        (void)hipEventDestroy(e);
}

extern "C" {

int rt_abi_version(void) { return RT_ABI_VERSION; }
int rt_sizeof_surface(void) { return (int)sizeof(rt_surface); }
int rt_sizeof_opd_args(void) { return (int)sizeof(rt_opd_args); }

int rt_device_count(int *count)
{
    if (!count)
        return rt_fail(NULL, RT_ERR_ARG, "rt_device_count: NULL");
    *count = 0;
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) {
        *count = 0;
        return rt_fail(NULL, RT_ERR_HIP, "hipGetDeviceCount: %s",
                       hipGetErrorString(e));
    }
    return RT_OK;
}

const char *rt_last_error(const rt_ctx *ctx) { return ctx ? ctx->err : g_err; }

int rt_create(int device, rt_ctx **out)
{
    if (!out)
        return rt_fail(NULL, RT_ERR_ARG, "rt_create: out is NULL");
    *out = NULL;
    int count = 0;
    int rc = rt_device_count(&count);
    if (rc != RT_OK)
        return rc;
    if (device < 0 || device >= count)
        return rt_fail(NULL, RT_ERR_ARG,
                       "rt_create: device %d not in [0,%d): no MI355X visible",
                       device, count);
    rt_ctx *c = (rt_ctx *)calloc(1, sizeof(rt_ctx));
    if (!c)
        return rt_fail(NULL, RT_ERR_NOMEM, "rt_create: host allocation");
    c->device = device;
    c->opt_alias = 1;
    c->opt_fuse = 1;
    c->opt_regen = 1;
    {
        /* even aspheres: the FMA / rcp / rsq Newton solve (rt_math.h), within
         * the 1e-8 contract of iterated aspheres, is the default;
         * RT_MI355_EXACT_ASPHERE=1 (or rt_set_option "exact_asphere") makes
         * the bit-for-bit restatement of scipy's iteration the default of
         * every context of the process */
        const char *e = getenv("RT_MI355_EXACT_ASPHERE");
        c->opt_fast = (e && atoi(e)) ? 0 : 1;
    }
    c->opt_resident = -1;
    c->opt_range = 1;
    c->opt_onepass = 1;
    {
        /* RT_MI355_BLOCK_RAYS=B: every batch of more than B rays is cut into
         * blocks of B (tests: the whole suite on a batch in blocks) */
        const char *e = getenv("RT_MI355_BLOCK_RAYS");
        c->opt_block = e ? atoi(e) : 0;
        if (c->opt_block < 0)
            c->opt_block = 0;
        /* RT_MI355_TURN_POINTS=P: generated batches of several bundles in
         * turns of P pupil points whatever their size (tests) */
        e = getenv("RT_MI355_TURN_POINTS");
        c->opt_turn = e ? atoll(e) : 0;
        if (c->opt_turn < -1)
            c->opt_turn = 0;
    }
    {
        const char *e = getenv("RT_MI355_PLACEMENT");
        c->opt_place = (e && !atoi(e)) ? 0 : 1;
        c->opt_place_good = RT_PLACE_GOOD_GBPS;
        c->opt_place_budget_ms = RT_PLACE_BUDGET_MS;
        c->opt_place_orders = -1;
    }
    c->opt_uniform = 1;
    c->opt_compact_every = 4; /* measured best, profiles/r02_probes */
#define RT_HIP_C(call)                                                        \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            rt_fail(NULL, RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
            rt_destroy(c); /* whatever exists so far */                       \
            return RT_ERR_HIP;                                                \
        }                                                                     \
    } while (0)
    RT_HIP_C(hipSetDevice(device));
    RT_HIP_C(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    RT_HIP_C(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
    RT_HIP_C(hipEventCreate(&c->k0));
    RT_HIP_C(hipEventCreate(&c->k1));
    for (int i = 0; i < RT_NEVENTS; ++i)
        RT_HIP_C(hipEventCreate(&c->ev[i]));
    for (int i = 0; i < RT_GATHER_SLOTS; ++i) {
        RT_HIP_C(hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming));
        RT_HIP_C(
            hipEventCreateWithFlags(&c->gathered[i], hipEventDisableTiming));
    }
    RT_HIP_C(hipEventCreate(&c->g0));
    RT_HIP_C(hipEventCreate(&c->g1));
    c->ngroups = 1;
    c->tab_cap = (size_t)4 * RT_MAX_SURFACES; /* grows in rt_upload_system */
    c->h_surf = (rt_surface *)calloc(c->tab_cap, sizeof(rt_surface));
    if (!c->h_surf) {
        rt_destroy(c);
        return rt_fail(NULL, RT_ERR_NOMEM, "rt_create: host allocation");
    }
    for (int k = 0; k < 2; ++k) {
        RT_HIP_C(hipMalloc((void **)&c->d_tab[k],
                           sizeof(rt_surface) * c->tab_cap));
        RT_HIP_C(hipHostMalloc((void **)&c->h_pinned[k],
                               sizeof(rt_surface) * c->tab_cap));
        RT_HIP_C(hipEventCreateWithFlags(&c->tab_used[k],
                                         hipEventDisableTiming));
    }
    c->d_surf = c->d_tab[0];
    c->h_stage = c->h_pinned[0];
    memset(c->keep, 1, sizeof c->keep);
#undef RT_HIP_C
    *out = c;
    return RT_OK;
}

int rt_destroy(rt_ctx *ctx)
{
    if (!ctx)
        return RT_OK;
    /* also the way out of a failed rt_create: every member is either what
     * calloc left (NULL) or a live resource */
    (void)hipSetDevice(ctx->device);
    if (ctx->stream)
        (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm_stream)
        (void)hipStreamSynchronize(ctx->comm_stream);
    rt_comm_destroy(ctx);
    if (ctx->d_buf)
        (void)rt_buf_free(ctx, ctx->d_buf);
    if (ctx->d_uni)
        (void)hipFree(ctx->d_uni);
    if (ctx->d_scratch)
        (void)hipFree(ctx->d_scratch);
    if (ctx->d_user)
        (void)hipFree(ctx->d_user);
    if (ctx->h_aim)
        (void)hipHostFree(ctx->h_aim);
    for (int i = 0; i < 2; ++i)
        if (ctx->h_pin[i]) {
            (void)hipHostFree(ctx->h_pin[i]);
            (void)hipEventDestroy(ctx->pin_done[i]);
        }
    if (ctx->d_w)
        (void)hipFree(ctx->d_w);
    if (ctx->d_partials)
        (void)hipFree(ctx->d_partials);
    if (ctx->h_res)
        (void)hipHostFree(ctx->h_res);
    if (ctx->h_group)
        (void)hipHostFree(ctx->h_group);
    if (ctx->seed_ev)
        (void)hipEventDestroy(ctx->seed_ev);
    if (ctx->h_rows)
        (void)hipHostFree(ctx->h_rows);
    if (ctx->d_arrived)
        (void)hipFree(ctx->d_arrived);
    if (ctx->d_group)
        (void)hipFree(ctx->d_group);
    if (ctx->d_gen)
        (void)hipFree(ctx->d_gen);
    if (ctx->d_opd_ref)
        (void)hipFree(ctx->d_opd_ref);
    if (ctx->d_opd)
        (void)hipFree(ctx->d_opd);
    if (ctx->h_opd)
        (void)hipHostFree(ctx->h_opd);
    for (int k = 0; k < 2; ++k) {
        if (ctx->d_tab[k])
            (void)hipFree(ctx->d_tab[k]);
        if (ctx->h_pinned[k])
            (void)hipHostFree(ctx->h_pinned[k]);
        rt_event_free(ctx->tab_used[k]);
    }
    free(ctx->h_surf);
    rt_comm_release(ctx);
    for (int i = 0; i < RT_GATHER_SLOTS; ++i) {
        rt_event_free(ctx->staged[i]);
        rt_event_free(ctx->gathered[i]);
    }
    rt_event_free(ctx->g0);
    rt_event_free(ctx->g1);
    for (int i = 0; i < RT_NEVENTS; ++i)
        rt_event_free(ctx->ev[i]);
    rt_event_free(ctx->k0);
    rt_event_free(ctx->k1);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    if (ctx->comm_stream)
        (void)hipStreamDestroy(ctx->comm_stream);
    if (ctx->copy_stream)
        (void)hipStreamDestroy(ctx->copy_stream);
    free(ctx);
    return RT_OK;
}

int rt_upload_system_groups(rt_ctx *ctx, const rt_surface *surf, int nsurf,
                            int ngroups)
{
    if (!ctx || !surf)
        return rt_fail(ctx, RT_ERR_ARG, "rt_upload_system: NULL argument");
    if (nsurf < 2 || nsurf > RT_MAX_SURFACES)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_upload_system: nsurf=%d not in [2,%d]", nsurf,
                       RT_MAX_SURFACES);
    if (ngroups < 1 || ngroups > RT_MAX_GROUPS)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_upload_system: ngroups=%d not in [1,%d]", ngroups,
                       RT_MAX_GROUPS);
    for (int j = 0; j < nsurf * ngroups; ++j) {
        if (surf[j].nasph < 0 || surf[j].nasph > RT_MAX_ASPH)
            return rt_fail(ctx, RT_ERR_ARG,
                           "rt_upload_system: element %d has %d aspheric "
                           "terms, limit %d",
                           j % nsurf, surf[j].nasph, RT_MAX_ASPH);
    }
    const size_t ntab = (size_t)nsurf * ngroups;
    if (ntab > ctx->tab_cap) {
        /* the device table may be in use by a kernel in flight, the pinned
         * one by a pending DMA */
        RT_HIP(ctx, hipSetDevice(ctx->device));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        rt_surface *h = (rt_surface *)calloc(ntab, sizeof(rt_surface));
        if (!h)
            return rt_fail(ctx, RT_ERR_NOMEM, "rt_upload_system: %zu tables",
                           (size_t)ngroups);
        free(ctx->h_surf);
        ctx->h_surf = h;
        ctx->tab_cap = 0;
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(ctx->d_tab[k]);
            (void)hipHostFree(ctx->h_pinned[k]);
            ctx->d_tab[k] = NULL;
            ctx->h_pinned[k] = NULL;
        }
        ctx->d_surf = NULL;
        ctx->h_stage = NULL;
        ctx->nsurf = 0; /* nothing traces until a table is in place again */
        for (int k = 0; k < 2; ++k) {
            RT_HIP(ctx, hipMalloc((void **)&ctx->d_tab[k],
                                  sizeof(rt_surface) * ntab));
            RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_pinned[k],
                                      sizeof(rt_surface) * ntab));
        }
        ctx->d_surf = ctx->d_tab[ctx->tab_cur];
        ctx->h_stage = ctx->h_pinned[ctx->tab_cur];
        ctx->tab_cap = ntab;
        ctx->table_dirty = 1;
    } else if (nsurf == ctx->nsurf && ngroups == ctx->ngroups &&
               !memcmp(ctx->h_surf, surf, sizeof(rt_surface) * ntab)) {
        /* the table the device already holds (a propagate() that re-packs an
         * unchanged System, rayopt/geometric_trace.py:98-99 allows edits
         * between calls): nothing to finalise, nothing to send */
        return RT_OK;
    }
    memcpy(ctx->h_surf, surf, sizeof(rt_surface) * ntab);
    /* the bits and the field the library fills in itself are the library's
     * everywhere -- h_surf is read as it is by rt_aim_pupil and by the
     * generation's first intercept, which never see the finalised table: a
     * caller's stray RT_F_RANGE there would send sphere intercepts through
     * rt_quot with rc = 0 */
    for (size_t j = 0; j < ntab; ++j) {
        ctx->h_surf[j].flags &= ~(RT_F_STORE_I | RT_F_NOSTORE | RT_F_SKIP_U |
                                  RT_F_FAST | RT_F_RANGE);
        ctx->h_surf[j].rc = 0.;
    }
    rt_pieces_reset(ctx); /* (an unchanged table returned above) */
    ctx->table_dirty = 1; /* finalised and sent by the next rt_trace */
    ctx->nsurf = nsurf;
    ctx->ngroups = ngroups;
    return RT_OK;
}

int rt_upload_system(rt_ctx *ctx, const rt_surface *surf, int nsurf)
{
    return rt_upload_system_groups(ctx, surf, nsurf, 1);
}

static_assert(RT_BLOCK == 256 && RT_CB == 256 && RT_LAY_WG == 256,
              "rt_block_plan cuts batches into whole 256-ray workgroups");

int rt_reserve(rt_ctx *ctx, int64_t nrays)
{
    if (!ctx || nrays < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_reserve: bad argument");
    if (ctx->nsurf < 2)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_reserve: rt_upload_system must come first");
    const int64_t quantum = 64;
    int64_t bs;
    int nblk;
    rt_block_plan(ctx->nsurf, ctx->opt_block, quantum, nrays, &bs, &nblk);
    if (ctx->opt_pitch > 0) {
        /* the row pitch (= rays per block) a multiple of opt_pitch rays */
        bs = (bs + ctx->opt_pitch - 1) / ctx->opt_pitch * ctx->opt_pitch;
        nblk = nblk > 1 ? (int)((nrays + bs - 1) / bs) : 1;
    }
    const int64_t ld = bs * nblk;
    rt_pieces_reset(ctx);
    ctx->opd_n = 0; /* path differences kept on the device: of the old batch */
    if (ld == ctx->ld && bs == ctx->bs && ctx->buf_nsurf == ctx->nsurf &&
        ctx->d_buf) {
        if (nrays != ctx->n) {
            ctx->gen_live = 0;
            ctx->uni_valid = 0;
        }
        ctx->n = nrays;
        return RT_OK;
    }
    ctx->gen_live = 0;
    ctx->uni_valid = 0;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t need = (size_t)ctx->nsurf * 10 * (size_t)ld;
    ctx->plan_L = ctx->nsurf; /* what the placement lays its pieces out for */
    ctx->plan_bs = bs;
    ctx->plan_nblk = nblk;
    const int vm_failures_before = g_place_vm_failures;
    const double t_place = rt_place_now_ms();
    bool allocated = false;
    ctx->place_deadline_ms = 0.; /* set by the allocation's first search */
    if (need > ctx->cap_doubles) {
        /* from here until the new buffer exists the context holds no rays:
         * a failed allocation must not leave the old sizes without a buffer */
        ctx->n = 0;
        ctx->ld = 0;
        ctx->bs = 0;
        ctx->nblk = 0;
        ctx->bts = 0;
        ctx->traced = 0;
        memset(ctx->valid, 0, sizeof ctx->valid);
        if (ctx->d_buf)
            RT_HIP(ctx, rt_buf_free(ctx, ctx->d_buf));
        ctx->d_buf = NULL;
        ctx->cap_doubles = 0;
        hipError_t e = rt_buf_alloc(ctx, (void **)&ctx->d_buf,
                                    need * sizeof(double));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return rt_fail(ctx, RT_ERR_NOMEM,
                           "rt_reserve: hipMalloc of %.3f GB failed: %s",
                           need * 8e-9, hipGetErrorString(e));
        }
        ctx->cap_doubles = need;
        allocated = true;
    }
    const size_t tiles = (size_t)(ld / 64);
    if (tiles > ctx->uni_cap) {
        if (ctx->d_uni)
            (void)hipFree(ctx->d_uni);
        ctx->d_uni = NULL;
        ctx->uni_cap = 0;
        /* note[tiles] (padded to 8 bytes) | first[6][tiles] */
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_uni,
                              rt_tiles_bytes(tiles)));
        ctx->uni_cap = tiles;
    }
    const bool fresh = ctx->ld != ld || ctx->bs != bs ||
                       ctx->buf_nsurf != ctx->nsurf;
    ctx->n = nrays;
    ctx->ld = ld;
    ctx->bs = bs;
    ctx->nblk = nblk;
    ctx->bts = nblk > 1 ? (int64_t)10 * ctx->nsurf * bs : 0;
    ctx->buf_nsurf = ctx->nsurf;
    ctx->traced = 0;
    /* placed arrays in a new layout: measure the store pattern over them
     * (nothing lives in the rows yet) */
    if (ctx->place.base && fresh && ctx->place.settled && !allocated) {
        /* a buffer that is being reused has had its search (a
         * caller alternating batch sizes paid for up to five sets of pieces
         * at every change): the new layout's pattern is measured over it --
         * that decides the resident workgroups per CU -- and nothing is
         * mapped, so there is nothing new to prove either */
        rt_place_tune(ctx, ctx->nsurf, ld);
    } else if (ctx->place.base && fresh) {
        if (!allocated) /* (an older buffer that never had its search) */
            ctx->place_deadline_ms = rt_place_now_ms() + ctx->opt_place_budget_ms;
        rt_place_settle(ctx, ctx->nsurf, ld, ctx->cap_doubles * sizeof(double));
        if (!ctx->d_buf) { /* (the mapping was lost on the way) */
            rt_place_release(&ctx->place);
            ctx->cap_doubles = 0;
            ctx->n = ctx->ld = ctx->bs = 0;
            ctx->nblk = 0;
            ctx->bts = 0;
            return rt_fail(ctx, RT_ERR_NOMEM,
                           "rt_reserve: the arrays could not be mapped");
        }
        if (!rt_place_coherent(ctx)) {
            /* kernels and copies do not see the same memory behind the range
             * (translations of an earlier mapping alive in the device, what
             * rt_place_flush is there to prevent): no results through THAT.
             * Plain allocations from here on, for every context of the
             * process */
            g_place_distrust = 1;
            ctx->place_incoherent = 1;
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            rt_place_release(&ctx->place);
            ctx->d_buf = NULL;
            const size_t bytes = ctx->cap_doubles * sizeof(double);
            ctx->cap_doubles = 0;
            hipError_t e = hipMalloc((void **)&ctx->d_buf, bytes);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                ctx->n = ctx->ld = ctx->bs = 0;
                ctx->nblk = 0;
                ctx->bts = 0;
                return rt_fail(ctx, RT_ERR_NOMEM,
                               "rt_reserve: hipMalloc of %.3f GB failed: %s",
                               bytes * 1e-9, hipGetErrorString(e));
            }
            ctx->cap_doubles = bytes / sizeof(double);
        }
        ctx->place.total_ms = (float)(rt_place_now_ms() - t_place);
    }
    ctx->place_deadline_ms = 0.;
    if (g_place_vm_failures != vm_failures_before)
        /* not fatal -- the arrays are there -- but never silent */
        (void)rt_fail(ctx, RT_OK,
                      "rt_reserve: %d virtual-memory call(s) failed while "
                      "giving memory back (first: %s)",
                      g_place_vm_failures - vm_failures_before,
                      g_place_vm_first);
    memset(ctx->i_alias, 0, sizeof ctx->i_alias);
    memset(ctx->u_alias, 0, sizeof ctx->u_alias);
    memset(ctx->valid, 0, sizeof ctx->valid);
    return RT_OK;
}

int64_t rt_nrays(const rt_ctx *ctx) { return ctx ? ctx->n : 0; }
int64_t rt_ld(const rt_ctx *ctx) { return ctx ? ctx->ld : 0; }

int rt_blocks(const rt_ctx *ctx, int64_t info[3])
{
    if (!ctx || !info)
        return RT_ERR_ARG;
    info[0] = ctx->nblk;
    info[1] = ctx->bs;
    info[2] = ctx->bts;
    return RT_OK;
}
int rt_nsurf(const rt_ctx *ctx) { return ctx ? ctx->nsurf : 0; }

int rt_need_scratch(rt_ctx *ctx, size_t bytes)
{
    if (bytes <= ctx->scratch_bytes)
        return RT_OK;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_scratch)
        RT_HIP(ctx, hipFree(ctx->d_scratch));
    ctx->d_scratch = NULL;
    ctx->scratch_bytes = 0;
    hipError_t e = hipMalloc(&ctx->d_scratch, bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return rt_fail(ctx, RT_ERR_NOMEM, "scratch hipMalloc(%zu): %s", bytes,
                       hipGetErrorString(e));
    }
    ctx->scratch_bytes = bytes;
    return RT_OK;
}

/* rays lo .. hi - 1 of row 0 from the staged launch rays (hi == n: the
 * padding columns up to ld too), on `stream` */
static int rt_seed_window(rt_ctx *ctx, const double *d_y, const double *d_u,
                          int64_t n, int layout, int64_t period, int64_t lo,
                          int64_t hi, hipStream_t stream)
{
    const int block = 256;
    const int64_t j1 = hi >= n ? ctx->ld : hi;
    const unsigned grid = (unsigned)((j1 - lo + block - 1) / block);
    if (layout == RT_LAYOUT_AOS)
        hipLaunchKernelGGL(rt_seed_aos_kernel, dim3(grid), dim3(block), 0,
                           stream, d_y, d_u, n, rt_layout(ctx), ctx->ld,
                           !ctx->opt_alias, period, rt_tiles_of(ctx, 0, true),
                           lo, j1);
    else
        hipLaunchKernelGGL(rt_seed_soa_kernel, dim3(grid), dim3(block), 0,
                           stream, d_y, d_u, n, rt_layout(ctx), ctx->ld,
                           !ctx->opt_alias, period, rt_tiles_of(ctx, 0, true),
                           lo, j1);
    RT_HIP(ctx, hipGetLastError());
    return RT_OK;
}

/* What share of the launch components is uniform across a tile, from 64
 * of the notes the seed kernel has just written (large batches only: the
 * answer picks the resident workgroups per CU, rt_resident_lds, which small
 * batches do not cap at all).  The stream has been synchronised. */
static int rt_sample_notes(rt_ctx *ctx)
{
    ctx->uni_share = -1.f;
    const int64_t tiles = ctx->ld / 64;
    if (!ctx->uni_valid || !ctx->opt_uniform || !ctx->d_uni ||
        ctx->n < ((int64_t)1 << 19) || tiles < 64)
        return RT_OK;
    unsigned notes[64];
    const size_t pitch = (size_t)(tiles / 64) * sizeof(unsigned);
    RT_HIP(ctx, hipMemcpy2D(notes, sizeof(unsigned), ctx->d_uni, pitch,
                            sizeof(unsigned), 64, hipMemcpyDeviceToHost));
    int bits = 0;
    for (int k = 0; k < 64; ++k)
        bits += __builtin_popcount(notes[k] & 63u);
    ctx->uni_share = (float)bits / (64.f * 6.f);
    return RT_OK;
}

static void rt_seed_begin(rt_ctx *ctx)
{
    ctx->uni_share = -1.f;
    ctx->uni_valid = 0;
    ctx->opd_n = 0; /* (kept path differences: of the rays that were here) */
}

static void rt_seed_done(rt_ctx *ctx)
{
    ctx->uni_valid = 1; /* the notes describe row 0 */
    ctx->i_alias[0] = ctx->opt_alias ? 2 : 0; /* i[0] = u[0] (:67) */
    ctx->u_alias[0] = 0;
    ctx->valid[0] = 1;
    ctx->gen_pending = 0; /* these rays replace a generated batch */
    ctx->gen_live = 0;
}

static int rt_seed(rt_ctx *ctx, const double *d_y, const double *d_u,
                   int64_t n, int layout, int64_t period)
{
    rt_seed_begin(ctx);
    const int rc = rt_seed_window(ctx, d_y, d_u, n, layout, period, 0, n,
                                  ctx->stream);
    if (rc != RT_OK)
        return rc;
    rt_seed_done(ctx);
    return RT_OK;
}

/* bytes of one staging buffer (RT_PIN_CHUNK_MIB: measurements) */
static size_t rt_pin_chunk(void)
{
    static size_t v = 0;
    if (!v) {
        const char *e = getenv("RT_PIN_CHUNK_MIB");
        const long m = e ? atol(e) : 0;
        v = (size_t)(m >= 4 && m <= 512 ? m : 32) << 20;
    }
    return v;
}
#define RT_PIN_CHUNK rt_pin_chunk()

/* the staging buffers are read and written by the host's copy threads:
 * RT_PIN_NONCOHERENT=1 asks for host-cached (non-coherent) pinned memory --
 * visibility is ordered by the event waits around every DMA */
static unsigned rt_pin_flags(void)
{
    const char *e = getenv("RT_PIN_NONCOHERENT");
    return e && atoi(e) ? hipHostMallocNonCoherent : hipHostMallocDefault;
}

/* what the device wrote into NON-coherent host memory is only guaranteed
 * visible to the host after waiting for an event that releases to the system
 * scope (the copy kernel writes the staging buffers and the copy
 * threads read them right after the wait) */
static unsigned rt_pin_event_flags(void)
{
    return rt_pin_flags() == hipHostMallocNonCoherent
               ? hipEventDisableTiming | hipEventReleaseToSystem
               : hipEventDisableTiming;
}

/*
 * memcpy between pageable memory and the pinned staging buffers on a few
 * threads: one core copies ~30 GB/s, the DMA engine moves ~55 GB/s over
 * PCIe 5 x16, so the single-threaded staging copy was the slower half of the
 * pipeline (RT_COPY_THREADS overrides the default of 8; 1 = plain memcpy).
 */
static int rt_copy_threads(void)
{
    static int n = 0;
    if (!n) {
        const char *e = getenv("RT_COPY_THREADS");
        n = e ? atoi(e) : 8;
        n = n < 1 ? 1 : (n > 16 ? 16 : n);
    }
    return n;
}

/* device -> pinned host by a KERNEL (the staging buffers are mapped into the
 * device's address space): 53 GB/s like a DMA at its best -- but the DMA of
 * hipMemcpyAsync has two levels on this platform, 55 and 24-26 GB/s for the
 * same 20 MB, and which one a process gets flips while it runs
 * (profiles/r05_final/d2h_*: a fresh stream of a fresh process at 0.88 ms,
 * the same stream at 0.38 after a context was created next to it; most boxes
 * gave the slow one: 10.4-10.7 ms per 240 MB row, 5.3 on two boxes) */
__global__ void __launch_bounds__(256)
rt_copy_out_kernel(const double *__restrict__ src, double *__restrict__ dst,
                   size_t n, int wide)
{
    const size_t step = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (wide) {
        const double2 *s2 = (const double2 *)src;
        double2 *d2 = (double2 *)dst;
        for (; i < n / 2; i += step)
            d2[i] = s2[i];
    } else {
        for (; i < n; i += step)
            dst[i] = src[i];
    }
}

static hipError_t rt_copy_by_kernel(hipStream_t s, void *dst, const void *src,
                                    size_t bytes)
{
    const int wide = bytes % 16 == 0 && (uintptr_t)src % 16 == 0 &&
                     (uintptr_t)dst % 16 == 0;
    hipLaunchKernelGGL(rt_copy_out_kernel, dim3(1024), dim3(256), 0, s,
                       (const double *)src, (double *)dst,
                       bytes / sizeof(double), wide);
    return hipGetLastError();
}

/*
 * Host -> device copy of a pageable buffer through two pinned staging
 * buffers: the CPU fills one while the DMA engine drains the other.  A plain
 * hipMemcpyAsync from pageable memory is staged by the runtime in small
 * pieces and reaches ~5 GB/s; this path is bound by the host memcpy.
 */
int rt_h2d(rt_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes < RT_PIN_CHUNK / 8) {
        RT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                                   ctx->stream));
        return RT_OK;
    }
    for (int i = 0; i < 2; ++i)
        if (!ctx->h_pin[i]) {
            RT_HIP(ctx, hipHostMalloc(&ctx->h_pin[i], RT_PIN_CHUNK,
                                      rt_pin_flags()));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_done[i],
                                                rt_pin_event_flags()));
        }
    /* (the buffers alternate ACROSS calls too: a windowed upload is a chain
     * of one-chunk calls) */
    int k = ctx->pin_next & 1;
    /* the first chunk is a quarter: nothing crosses PCIe while it is staged */
    size_t len = 0;
    for (size_t off = 0; off < bytes; off += len, k ^= 1, ctx->pin_next = k) {
        const size_t chunk = off || bytes <= RT_PIN_CHUNK ? RT_PIN_CHUNK
                                                          : RT_PIN_CHUNK / 4;
        len = bytes - off < chunk ? bytes - off : chunk;
        if (ctx->pin_busy[k]) /* this call's or an earlier call's DMA */
            RT_HIP(ctx, hipEventSynchronize(ctx->pin_done[k]));
        rt_memcpy_mt(ctx->h_pin[k], (const char *)src + off, len,
                     rt_copy_threads());
        /* RT_H2D_KERNEL=1 (A/B): a copy kernel reading the mapped staging
         * buffer instead of the DMA engine */
        static const bool kern = getenv("RT_H2D_KERNEL") != NULL;
        if (kern && len % sizeof(double) == 0)
            RT_HIP(ctx, rt_copy_by_kernel(ctx->stream, (char *)dst + off,
                                          ctx->h_pin[k], len));
        else
            RT_HIP(ctx, hipMemcpyAsync((char *)dst + off, ctx->h_pin[k], len,
                                       hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->pin_done[k], ctx->stream));
        ctx->pin_busy[k] = 1;
    }
    return RT_OK;
}

/* device -> pageable host through the same two staging buffers: a list of
 * copies as ONE pipeline (the rows of a download, block segment by block
 * segment: the buffers stay busy across the borders between them);
 * synchronous */
struct rt_copy_job {
    void *dst;
    const void *src;
    size_t bytes;
};

static hipError_t rt_copy_out(hipStream_t s, void *dst, const void *src,
                              size_t bytes)
{
    static const bool dma = getenv("RT_D2H_DMA") != NULL; /* A/B */
    if (dma || bytes % sizeof(double))
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s);
    return rt_copy_by_kernel(s, dst, src, bytes);
}

static int rt_d2h_jobs(rt_ctx *ctx, const rt_copy_job *jobs, size_t njobs)
{
    for (int i = 0; i < 2; ++i)
        if (!ctx->h_pin[i]) {
            RT_HIP(ctx, hipHostMalloc(&ctx->h_pin[i], RT_PIN_CHUNK,
                                      rt_pin_flags()));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_done[i],
                                                rt_pin_event_flags()));
        }
    /* the copies run on a stream of their own, behind everything the trace
     * stream has been given so far */
    if (!ctx->copy_stream)
        RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream,
                                             hipStreamNonBlocking));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const hipStream_t cs = ctx->copy_stream;
    /* the chunks of all jobs in order, at most RT_PIN_CHUNK each */
    size_t j = 0, off = 0;
    auto next = [&](void **dst, const void **src, size_t *len) {
        while (j < njobs && off >= jobs[j].bytes) {
            ++j;
            off = 0;
        }
        if (j >= njobs)
            return false;
        /* a job in equal chunks (40 MB: 20 + 20, not 32 + 8: the short
         * ones cost a DMA call and a team of threads like the long ones) */
        const size_t pieces = (jobs[j].bytes + RT_PIN_CHUNK - 1) / RT_PIN_CHUNK;
        size_t even = ((jobs[j].bytes + pieces - 1) / pieces + 4095) &
                      ~(size_t)4095;
        /* the very first chunk is short: no host thread copies while it
         * crosses PCIe */
        if (j == 0 && off == 0 && even > RT_PIN_CHUNK / 4)
            even = RT_PIN_CHUNK / 4;
        *len = jobs[j].bytes - off < even ? jobs[j].bytes - off : even;
        *dst = (char *)jobs[j].dst + off;
        *src = (const char *)jobs[j].src + off;
        off += *len;
        return true;
    };
    /* chunk i-1 leaves its staging buffer on the copy threads -- all of it:
     * the caller only waits -- while the copy kernel of chunk i fills the
     * other one: per 20 MB chunk 0.39 ms of kernel beside 0.3-0.5 ms of
     * memcpy, 5.6-5.7 ms per 240 MB row (42 GB/s; the caller's own share of
     * every chunk, copied after its wait, and the DMA's slow level made that
     * 10.4-10.7 ms on most boxes until round 5) */
    rt_copy_team team;
    void *prev_dst = NULL;
    size_t prev_len = 0;
    bool have_prev = false;
    for (size_t i = 0;; ++i) {
        void *dst = NULL;
        const void *src = NULL;
        size_t len = 0;
        const bool more = next(&dst, &src, &len);
        if (have_prev) { /* chunk i-1 has landed: start draining it */
            hipError_t e = hipEventSynchronize(ctx->pin_done[(i - 1) & 1]);
            if (e != hipSuccess)
                return rt_fail(ctx, RT_ERR_HIP, "rt_d2h: %s",
                               hipGetErrorString(e));
            rt_copy_start(&team, prev_dst, ctx->h_pin[(i - 1) & 1], prev_len,
                          rt_copy_threads(), more);
        }
        hipError_t e = hipSuccess;
        if (more) { /* chunk i into the other buffer */
            if (ctx->pin_busy[i & 1]) /* an upload may still read it */
                e = hipEventSynchronize(ctx->pin_done[i & 1]);
            if (e == hipSuccess)
                e = rt_copy_out(cs, ctx->h_pin[i & 1], src, len);
            if (e == hipSuccess)
                e = hipEventRecord(ctx->pin_done[i & 1], cs);
        }
        if (have_prev) { /* the threads are joined whatever the copy said */
            rt_copy_finish(&team);
            ctx->pin_busy[(i - 1) & 1] = 0;
        }
        if (e != hipSuccess)
            return rt_fail(ctx, RT_ERR_HIP, "rt_d2h: %s",
                           hipGetErrorString(e));
        if (!more)
            break;
        prev_dst = dst;
        prev_len = len;
        have_prev = true;
    }
    return RT_OK;
}

int rt_d2h(rt_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes < RT_PIN_CHUNK / 8) {
        RT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                                   ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    const rt_copy_job job = {dst, src, bytes};
    return rt_d2h_jobs(ctx, &job, 1);
}

/* nrow rows of ld doubles -> compact rows of n doubles on the host */
static int rt_rows_to_host(rt_ctx *ctx, double *dst, const double *src,
                           size_t nrow)
{
    const size_t rb = (size_t)ctx->n * sizeof(double);
    if (ctx->ld == ctx->n && ctx->nblk == 1) /* no padding: one contiguous block */
        return rt_d2h(ctx, dst, src, rb * nrow);
    /* segment by segment (one per block of the batch): the large ones as one
     * pipeline over all (row, segment) pairs, the small ones as 2-D copies */
    size_t nbig = 0;
    RT_FOR_SEGMENTS(ctx, g, 0, ctx->n)
        nbig += (size_t)g.cnt * sizeof(double) >= RT_PIN_CHUNK / 8 ? nrow : 0;
    rt_copy_job *jobs = nbig ? (rt_copy_job *)malloc(nbig * sizeof *jobs)
                             : NULL;
    if (nbig && !jobs)
        return rt_fail(ctx, RT_ERR_NOMEM, "rt_download: %zu copies", nbig);
    size_t k = 0;
    for (size_t r = 0; r < nrow; ++r) { /* row after row, as the host reads */
        RT_FOR_SEGMENTS(ctx, g, 0, ctx->n) {
            const size_t sb = (size_t)g.cnt * sizeof(double);
            if (sb >= RT_PIN_CHUNK / 8) {
                jobs[k].dst = dst + r * ctx->n + g.ray;
                jobs[k].src = src + r * ctx->bs + g.off;
                jobs[k++].bytes = sb;
            }
        }
    }
    hipError_t e = hipSuccess;
    RT_FOR_SEGMENTS(ctx, g, 0, ctx->n) {
        const size_t sb = (size_t)g.cnt * sizeof(double);
        if (sb < RT_PIN_CHUNK / 8 && e == hipSuccess)
            e = hipMemcpy2DAsync(dst + g.ray, rb, src + g.off,
                                 ctx->bs * sizeof(double), sb, nrow,
                                 hipMemcpyDeviceToHost, ctx->stream);
    }
    int rc = RT_OK;
    if (e == hipSuccess && nbig)
        rc = rt_d2h_jobs(ctx, jobs, nbig);
    free(jobs);
    if (e != hipSuccess)
        return rt_fail(ctx, RT_ERR_HIP, "rt_download: %s",
                       hipGetErrorString(e));
    if (rc != RT_OK)
        return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_set_rays_repeat(rt_ctx *ctx, const double *y, const double *u,
                       int64_t p, int copies, int layout)
{
    if (!ctx || !y || !u || p < 1 || copies < 1 ||
        (layout != RT_LAYOUT_AOS && layout != RT_LAYOUT_SOA))
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_rays: bad argument");
    const int64_t n = p * copies;
    int rc = rt_reserve(ctx, n);
    if (rc != RT_OK)
        return rc;
    const size_t bytes = (size_t)p * 3 * sizeof(double);
    rc = rt_need_scratch(ctx, 2 * bytes);
    if (rc != RT_OK)
        return rc;
    double *sy = (double *)ctx->d_scratch;
    double *su = sy + (size_t)p * 3;
    if (layout == RT_LAYOUT_AOS && copies == 1 && bytes > 2 * RT_PIN_CHUNK) {
        /* a large (n, 3) upload is seeded WINDOW BY WINDOW: y and u of
         * ~1.4 * 10^6 rays (one staging chunk each) cross PCIe, then their
         * seed kernel runs on a second stream while the next window's chunks
         * follow on the first -- the seed (0.25 ms at 10^7 rays) and the
         * pipeline's drain no longer stand behind the last byte */
        if (!ctx->copy_stream)
            RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream,
                                                 hipStreamNonBlocking));
        if (!ctx->seed_ev)
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->seed_ev,
                                                hipEventDisableTiming));
        const int64_t W = (int64_t)(RT_PIN_CHUNK / 24) / 256 * 256;
        rt_seed_begin(ctx);
        /* (everything queued on the trace stream so far comes first) */
        RT_HIP(ctx, hipEventRecord(ctx->seed_ev, ctx->stream));
        RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->seed_ev, 0));
        for (int64_t lo = 0; lo < n;) {
            const int64_t w = lo ? W : W / 4; /* a short first window */
            const int64_t hi = lo + w < n ? lo + w : n;
            const size_t wb = (size_t)(hi - lo) * 3 * sizeof(double);
            rc = rt_h2d(ctx, sy + lo * 3, y + lo * 3, wb);
            if (rc == RT_OK)
                rc = rt_h2d(ctx, su + lo * 3, u + lo * 3, wb);
            if (rc != RT_OK)
                return rc;
            RT_HIP(ctx, hipEventRecord(ctx->seed_ev, ctx->stream));
            RT_HIP(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->seed_ev, 0));
            rc = rt_seed_window(ctx, sy, su, n, layout, p, lo, hi,
                                ctx->copy_stream);
            if (rc != RT_OK)
                return rc;
            lo = hi;
        }
        RT_HIP(ctx, hipEventRecord(ctx->seed_ev, ctx->copy_stream));
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->seed_ev, 0));
        rt_seed_done(ctx);
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return rt_sample_notes(ctx);
    }
    rc = rt_h2d(ctx, sy, y, bytes);
    if (rc == RT_OK)
        rc = rt_h2d(ctx, su, u, bytes);
    if (rc != RT_OK)
        return rc;
    rc = rt_seed(ctx, sy, su, n, layout, p);
    if (rc != RT_OK)
        return rc;
    /* caller's host arrays may be released as soon as we return */
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return rt_sample_notes(ctx);
}

int rt_set_rays(rt_ctx *ctx, const double *y, const double *u, int64_t n,
                int layout)
{
    return rt_set_rays_repeat(ctx, y, u, n, 1, layout);
}

int rt_set_rays_device(rt_ctx *ctx, const double *d_y, const double *d_u,
                       int64_t n, int layout)
{
    if (!ctx || !d_y || !d_u || n < 1 ||
        (layout != RT_LAYOUT_AOS && layout != RT_LAYOUT_SOA))
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_rays_device: bad argument");
    int rc = rt_reserve(ctx, n);
    if (rc != RT_OK)
        return rc;
    return rt_seed(ctx, d_y, d_u, n, layout, n);
}


int rt_sizeof_field(void) { return (int)sizeof(rt_field); }

int rt_generate_rays(rt_ctx *ctx, const rt_field *fields, int nfields,
                     const double *pupil_xy, int64_t npupil)
{
    if (!ctx || !fields || !pupil_xy || nfields < 1 || npupil < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_generate_rays: bad argument");
    if (ctx->nsurf < 2)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_generate_rays: rt_upload_system must come first");
    const int64_t n = (int64_t)nfields * npupil;
    /* whatever happens below, row 0 is no longer what the previous frames
     * describe: set again once both copies are on their way */
    ctx->gen_live = 0;
    ctx->gen_pending = 0;
    ctx->uni_valid = 0;
    int rc = rt_reserve(ctx, n);
    if (rc != RT_OK)
        return rc;
    const size_t fbytes = sizeof(rt_field) * (size_t)nfields;
    const size_t fpad = (fbytes + 255) / 256 * 256;
    const size_t pbytes = sizeof(double) * 2 * (size_t)npupil;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (fpad + pbytes > ctx->gen_bytes) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->d_gen)
            (void)hipFree(ctx->d_gen);
        ctx->d_gen = NULL;
        ctx->gen_bytes = 0;
        RT_HIP(ctx, hipMalloc(&ctx->d_gen, fpad + pbytes));
        ctx->gen_bytes = fpad + pbytes;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_gen, fields, fbytes,
                               hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipMemcpyAsync((char *)ctx->d_gen + fpad, pupil_xy, pbytes,
                               hipMemcpyHostToDevice, ctx->stream));
    ctx->gen_fpad = fpad;
    ctx->gen_nf = nfields;
    ctx->gen_np = npupil;
    ctx->gen_n = n;
    ctx->gen_s0 = ctx->h_surf[0];
    /* row 0 is built by the first trace (rt_trace_gen_kernel), or by
     * rt_gen_flush as soon as anything else asks for it */
    ctx->gen_pending = 1;
    ctx->opd_n = 0;
    ctx->gen_live = 1;
    ctx->traced = 1;
    ctx->i_alias[0] = ctx->opt_alias ? 2 : 0;
    ctx->u_alias[0] = 0;
    ctx->valid[0] = 1;
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    if (!ctx->opt_fuse) {
        rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream)); /* ~0 when deferred */
    /* caller's host arrays may be released as soon as we return */
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_upload_row(rt_ctx *ctx, int which, int surf, const double *src_soa)
{
    if (!ctx || !src_soa || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_upload_row: bad argument");
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_upload_row: no such row %d",
                       surf);
    const int nc = rt_ncomp(which);
    {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    /* rows served from the one being replaced keep what they show now */
    if (which == RT_U && surf + 1 < ctx->buf_nsurf && ctx->valid[surf + 1] &&
        ctx->i_alias[surf + 1] == 1) {
        int rc = rt_detach(ctx, RT_I, surf + 1);
        if (rc != RT_OK)
            return rc;
    }
    if (which == RT_U && ctx->i_alias[surf] == 2) {
        /* i[0] is u[0] at seeding time (geometric_trace.py:67), a copy in
         * the reference: it keeps the old directions */
        int rc = rt_detach(ctx, RT_I, surf);
        if (rc != RT_OK)
            return rc;
    }
    if (which == RT_I && ctx->u_alias[surf]) {
        int rc = rt_detach(ctx, RT_U, surf);
        if (rc != RT_OK)
            return rc;
    }
    if (surf == 0 && which != RT_I && which != RT_T) {
        ctx->gen_live = 0; /* the launch rays are the caller's from here on */
        ctx->uni_valid = 0;
    }
    if (which == RT_I)
        ctx->i_alias[surf] = 0; /* now holds its own data */
    if (which == RT_U)
        ctx->u_alias[surf] = 0;
    ctx->valid[surf] = 1;
    rt_pieces_reset(ctx); /* a step in pieces does not survive a new row */
    double *dst = rt_row(ctx, which, surf);
    RT_FOR_SEGMENTS(ctx, g, 0, ctx->n)
        RT_HIP(ctx, hipMemcpy2DAsync(dst + g.off, ctx->bs * sizeof(double),
                                     src_soa + g.ray, ctx->n * sizeof(double),
                                     (size_t)g.cnt * sizeof(double), nc,
                                     hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_chunk_bounds(int64_t n, int chunk, int nchunks, int64_t *lo,
                    int64_t *hi)
{
    if (n < 0 || nchunks < 1 || chunk < 0 || chunk >= nchunks || !lo || !hi)
        return rt_fail(NULL, RT_ERR_ARG, "rt_chunk_bounds: chunk %d of %d",
                       chunk, nchunks);
    /* equal pieces of whole 256-ray workgroups; the last ones may be short
     * or empty */
    int64_t per = (n + nchunks - 1) / nchunks;
    per = (per + RT_BLOCK - 1) / RT_BLOCK * RT_BLOCK;
    *lo = (int64_t)chunk * per < n ? (int64_t)chunk * per : n;
    *hi = *lo + per < n ? *lo + per : n;
    return RT_OK;
}

/* the trace of the ray window [lo, hi) (whole batch: 0, n) */
static int rt_trace_window(rt_ctx *ctx, int start, int stop, int clip,
                           int64_t lo, int64_t hi)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_trace: NULL context");
    if (ctx->nsurf < 2 || !ctx->d_buf || ctx->n < 1)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_trace: upload a system and rays first");
    if (ctx->buf_nsurf != ctx->nsurf)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_trace: system length changed (%d -> %d) after the "
                       "rays were set",
                       ctx->buf_nsurf, ctx->nsurf);
    if (stop <= 0 || stop > ctx->nsurf)
        stop = ctx->nsurf;
    if (start < 1 || start > stop)
        return rt_fail(ctx, RT_ERR_ARG, "rt_trace: start=%d stop=%d nsurf=%d",
                       start, stop, ctx->nsurf);
    if (!ctx->valid[start - 1])
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_trace: seed row %d holds no data (not stored by "
                       "the previous trace)", start - 1);
    ctx->opd_n = 0; /* rt_opd_device hands out nothing of the rows before
                       this trace */
    if (ctx->ngroups > 1 && (ctx->n % ctx->ngroups ||
                             (ctx->n / ctx->ngroups) % 64))
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_trace: %lld rays do not split into %d groups of a "
                       "multiple of %d rays", (long long)ctx->n, ctx->ngroups,
                       64);
    const bool windowed = !(lo == 0 && hi == ctx->n);
    if (windowed && ctx->ngroups > 1)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_trace_chunk: not with ray groups (several surface "
                       "tables in one batch)");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->table_clip != (clip != 0))
        ctx->table_dirty = 1; /* what is stored depends on clip (SKIP_U) */
    /* the kernel reads the seed rows at their natural address, and rows
     * beyond `stop` that are served from a row about to be rewritten keep
     * what they show now */
    {
        int rc = rt_detach(ctx, RT_U, start - 1);
        if (rc == RT_OK && stop < ctx->buf_nsurf && ctx->valid[stop] &&
            ctx->i_alias[stop] == 1)
            rc = rt_detach(ctx, RT_I, stop);
        if (rc != RT_OK)
            return rc;
    }
    if (ctx->table_dirty) {
        /* the table goes into the buffer that is NOT in use: a kernel in
         * flight keeps reading the previous one, so back-to-back propagate()
         * calls on a changing System pipeline instead of draining the stream.
         * The other buffer was retired two uploads ago; tab_used[] marks the
         * last work that touched it */
        const int k = ctx->tab_cur ^ 1;
        RT_HIP(ctx, hipEventRecord(ctx->tab_used[ctx->tab_cur], ctx->stream));
        RT_HIP(ctx, hipEventSynchronize(ctx->tab_used[k]));
        ctx->tab_cur = k;
        ctx->d_surf = ctx->d_tab[k];
        ctx->h_stage = ctx->h_pinned[k];
        const int ntab = ctx->nsurf * ctx->ngroups;
        memcpy(ctx->h_stage, ctx->h_surf, sizeof(rt_surface) * ntab);
        /* whether a row is served from another one is decided per ROW, for
         * all tables at once: u[j] == i[j] bit for bit where no table bends
         * the ray at j and nothing clips it; i[j] == u[j-1] where no table
         * tilts element j or j-1 */
        unsigned char bends[RT_MAX_SURFACES] = {0};
        unsigned char tilted[RT_MAX_SURFACES] = {0};
        for (int jj = 0; jj < ntab; ++jj) {
            if (ctx->h_surf[jj].flags & RT_F_REFRACT)
                bends[jj % ctx->nsurf] = 1;
            if (ctx->h_surf[jj].flags & RT_F_ROTATED)
                tilted[jj % ctx->nsurf] = 1;
        }
        ctx->table_asph = ctx->gen_s0.flags & RT_F_ASPH ? 1 : 0;
        for (int jj = 0; jj < ntab; ++jj) {
            if (ctx->h_surf[jj].flags & RT_F_ASPH)
                ctx->table_asph = 1; /* (else: the kernels without Newton) */
            const int j = jj % ctx->nsurf; /* element index in its group */
            unsigned f = ctx->h_stage[jj].flags &
                         ~(RT_F_STORE_I | RT_F_NOSTORE | RT_F_SKIP_U);
            if (ctx->opt_alias && !clip && !bends[j] && j > 0)
                f |= RT_F_SKIP_U;
            if (!ctx->keep[j])
                f |= RT_F_NOSTORE;
            /* i[j] can only be served from U[j-1] if that row exists */
            const bool rot = tilted[j] || (j > 0 && tilted[j - 1]);
            const bool prev_kept =
                j > 0 && (j - 1 < start ? ctx->valid[j - 1] : ctx->keep[j - 1]);
            if (!ctx->opt_alias || rot || j == 0 || !prev_kept)
                f |= RT_F_STORE_I;
            f &= ~RT_F_FAST;
            if (ctx->opt_fast && (f & RT_F_ASPH)) {
                f |= RT_F_FAST;
                /* the fast path evaluates a fixed number of terms */
                rt_surface *S = ctx->h_stage + jj;
                for (int q = S->nasph; q < RT_MAX_ASPH; ++q)
                    S->asph[q] = S->dasph[q] = 0.;
            }
            /* IEEE quotients / square roots without their range
             * scaffolding (rt_math.h): the element's own operands must be
             * inside the range the per-ray checks assume */
            f &= ~RT_F_RANGE;
            if (ctx->opt_range) {
                const rt_surface *S = ctx->h_stage + jj;
                const bool sphere = (f & RT_F_CURVED) &&
                                    !(f & (RT_F_CONIC | RT_F_ASPH));
                const bool snell = (f & RT_F_REFRACT) && !(f & RT_F_MIRROR);
                if ((!sphere || rt_in_range(S->c)) &&
                    (!snell || rt_in_range(S->mu2m1)) &&
                    (!(f & RT_F_REFRACT) || rt_in_range(S->muf)))
                    f |= RT_F_RANGE;
            }
            ctx->h_stage[jj].flags = f;
            ctx->h_stage[jj].rc = 0.;
        }
        RT_HIP(ctx, hipMemcpyAsync(ctx->d_surf, ctx->h_stage,
                                   sizeof(rt_surface) * ntab,
                                   hipMemcpyHostToDevice, ctx->stream));
        /* rt_surface.rc: 1/c as the device's division refines it */
        hipLaunchKernelGGL(rt_table_finish_kernel, dim3((ntab + 63) / 64),
                           dim3(64), 0, ctx->stream, ctx->d_surf, ntab);
        RT_HIP(ctx, hipGetLastError());
        ctx->table_dirty = 0;
        ctx->table_start = start;
        ctx->table_clip = clip != 0;
    } else if (ctx->table_start != start) {
        ctx->table_dirty = 1; /* alias decisions depend on start */
        return rt_trace_window(ctx, start, stop, clip, lo, hi);
    }
    /* a generated batch that no one has looked at yet is built inside this
     * launch (from the first element on) */
    const bool gen_kernel = start == 1 && start < stop &&
                            !rt_use_compact(ctx, start, stop);
    const bool fused = ctx->gen_pending && gen_kernel && !windowed;
    /* a later trace of the same generated batch builds the rays again in
     * registers (same frames, same arithmetic: the values row 0 holds)
     * rather than read 48 B per ray among the saturated stores */
    const bool regen = (!ctx->gen_pending || windowed) && ctx->gen_live &&
                       ctx->opt_regen && ctx->opt_fuse && gen_kernel &&
                       ctx->valid[0];
    if (!fused) {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    if (lo == 0)
        RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    ctx->last_compact = 0;
    /* the window as the kernels see it: rays lo + (0 .. cols) (the last
     * window takes the padding slots up to ld); the arrays stay where they
     * are, rt_col adds j0 (a window may begin anywhere in any block) */
    rt_lay lay = rt_layout(ctx);
    rt_lay_set_window(lay, lo, ctx->ld);
    /* an empty trailing piece (lo == hi == n) launches nothing: the padding
     * columns belong to the last NON-empty piece */
    const int64_t cols = lo < hi ? (hi == ctx->n ? ctx->ld : hi) - lo : 0;
    const int64_t group_rays = ctx->ngroups > 1 ? ctx->n / ctx->ngroups : 0;
    const unsigned grid = (unsigned)((cols + RT_BLOCK - 1) / RT_BLOCK);
    const size_t lds = rt_resident_lds(ctx, start, stop);
    if (cols <= 0 || start >= stop) {
        /* an empty window, or nothing to trace */
    } else if (fused || regen) {
        ctx->gen_pending = 0;
        if (fused)
            ctx->uni_valid = 0; /* this launch writes row 0 */
        /* bundles over a large pupil: in turns (whole batch, bundles of
         * whole workgroups only; elsewhere the order stays the rays') */
        rt_gen_order order = {0, 0, 0};
        int64_t turn = ctx->opt_turn > 0 ? ctx->opt_turn
                       : 16. * ctx->gen_np > RT_TURN_ABOVE ? RT_TURN_POINTS
                                                           : 0;
        turn = turn / RT_BLOCK * RT_BLOCK;
        if (ctx->opt_turn >= 0 && turn > 0 && !windowed && lo == 0 &&
            ctx->gen_np % RT_BLOCK == 0 && ctx->gen_np > turn &&
            ctx->gen_n % ctx->gen_np == 0 && ctx->gen_n > ctx->gen_np &&
            ctx->gen_n / RT_BLOCK <= grid) {
            order.turn = (uint32_t)(turn / RT_BLOCK);
            order.per = (uint32_t)(ctx->gen_np / RT_BLOCK);
            order.nf = (uint32_t)(ctx->gen_n / ctx->gen_np);
        }
        const bool asph = ctx->table_asph || (ctx->gen_s0.flags & RT_F_ASPH);
        hipLaunchKernelGGL(asph ? rt_trace_gen_kernel<true>
                                : rt_trace_gen_kernel<false>,
                           dim3(grid), dim3(RT_BLOCK),
                           lds, ctx->stream, ctx->d_surf, stop, clip, lay, cols,
                           group_rays, ctx->nsurf, ctx->ngroups,
                           (const rt_field *)ctx->d_gen,
                           (const double *)((char *)ctx->d_gen + ctx->gen_fpad),
                           ctx->gen_np, ctx->gen_n, lo, ctx->gen_s0,
                           !ctx->opt_alias, fused ? 1 : 0, order);
        RT_HIP(ctx, hipGetLastError());
    } else if (!windowed && rt_use_compact(ctx, start, stop)) {
        hipLaunchKernelGGL(rt_trace_compact_kernel,
                           dim3((unsigned)((ctx->ld + RT_CB - 1) / RT_CB)),
                           dim3(RT_CB), 0, ctx->stream, ctx->d_surf, start,
                           stop, clip, lay, ctx->ld, group_rays, ctx->nsurf,
                           ctx->ngroups, ctx->opt_compact_every);
        RT_HIP(ctx, hipGetLastError());
        ctx->last_compact = 1;
    } else {
        /* launch components that are uniform across a 64-ray tile are
         * fetched once per wavefront (the seed kernels' notes on row 0) */
        const rt_tiles tiles = rt_tiles_of(
            ctx, lo / 64, start == 1 && ctx->uni_valid && ctx->opt_uniform);
        hipLaunchKernelGGL(ctx->table_asph ? rt_trace_kernel<true>
                                           : rt_trace_kernel<false>,
                           dim3(grid), dim3(RT_BLOCK), lds,
                           ctx->stream, ctx->d_surf, start, stop, clip, lay,
                           cols, group_rays, ctx->nsurf, ctx->ngroups, tiles);
        RT_HIP(ctx, hipGetLastError());
    }
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    for (int sidx = start; sidx < stop; ++sidx) {
        const unsigned f = ctx->h_stage[sidx].flags;
        ctx->valid[sidx] = !(f & RT_F_NOSTORE);
        ctx->i_alias[sidx] = (f & (RT_F_STORE_I | RT_F_NOSTORE)) ? 0 : 1;
        ctx->u_alias[sidx] = (f & RT_F_SKIP_U) && !(f & RT_F_NOSTORE);
    }
    ctx->traced = 1;
    return RT_OK;
}

/*
 * How many of the lanes a wavefront drags through the asphere iteration are
 * still iterating (north_star: "wavefront ballots for the asphere
 * iteration"; rayopt/elements.py:333-349 solves ray by ray): the batch is
 * marched again from row 0 by a census kernel that stores nothing.  Needs a
 * completed trace from element 1 (the table as that trace finalised it, row
 * 0 in place).
 */
int rt_newton_census(rt_ctx *ctx, int clip, uint64_t out[4])
{
    if (!ctx || !out)
        return rt_fail(ctx, RT_ERR_ARG, "rt_newton_census: NULL argument");
    RT_ROWS_WHOLE(ctx, "rt_newton_census");
    if (!ctx->d_buf || ctx->n < 1 || !ctx->traced || ctx->table_dirty ||
        ctx->table_start != 1 || !ctx->valid[0])
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_newton_census: trace the batch from element 1 "
                       "first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rt_gen_flush(ctx);
    if (rc == RT_OK)
        rc = rt_detach(ctx, RT_U, 0);
    if (rc == RT_OK)
        rc = rt_need_scratch(ctx, 4 * sizeof(unsigned long long));
    if (rc != RT_OK)
        return rc;
    unsigned long long *d = (unsigned long long *)ctx->d_scratch;
    RT_HIP(ctx, hipMemsetAsync(d, 0, 4 * sizeof *d, ctx->stream));
    rt_lay lay = rt_layout(ctx);
    rt_lay_set_window(lay, 0, ctx->ld);
    const int64_t group_rays = ctx->ngroups > 1 ? ctx->n / ctx->ngroups : 0;
    const unsigned grid = (unsigned)((ctx->ld + RT_BLOCK - 1) / RT_BLOCK);
    hipLaunchKernelGGL(rt_census_kernel, dim3(grid), dim3(RT_BLOCK), 0,
                       ctx->stream, ctx->d_surf, 1, ctx->nsurf, clip, lay,
                       ctx->ld, ctx->n, group_rays, ctx->nsurf, ctx->ngroups,
                       d);
    RT_HIP(ctx, hipGetLastError());
    unsigned long long h[4];
    RT_HIP(ctx, hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost,
                               ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < 4; ++k)
        out[k] = h[k];
    return RT_OK;
}

int rt_trace(rt_ctx *ctx, int start, int stop, int clip)
{
    const int rc = rt_trace_window(ctx, start, stop, clip, 0, ctx ? ctx->n : 0);
    if (ctx && rc == RT_OK)
        ctx->pieces_seen = ctx->pieces_total = 0; /* whole rows again */
    return rc;
}

int rt_trace_chunk(rt_ctx *ctx, int start, int stop, int clip, int chunk,
                   int nchunks)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_trace_chunk: NULL context");
    int64_t lo, hi;
    int rc = rt_chunk_bounds(ctx->n, chunk, nchunks, &lo, &hi);
    if (rc != RT_OK)
        return rt_fail(ctx, rc, "rt_trace_chunk: chunk %d of %d", chunk,
                       nchunks);
    rc = rt_trace_window(ctx, start, stop, clip, lo, hi);
    if (rc != RT_OK || nchunks == 1 || nchunks > 256) {
        if (rc == RT_OK)
            ctx->pieces_seen = ctx->pieces_total = 0;
        return rc;
    }
    /* which pieces of this step exist by now (any order) */
    if (ctx->pieces_seen == 0 || ctx->pieces_total != nchunks ||
        ctx->pieces_start != start || ctx->pieces_stop != stop ||
        ctx->pieces_clip != clip) {
        memset(ctx->pieces_mask, 0, sizeof ctx->pieces_mask);
        ctx->pieces_seen = 0;
        ctx->pieces_total = nchunks;
        ctx->pieces_start = start;
        ctx->pieces_stop = stop;
        ctx->pieces_clip = clip;
    }
    uint64_t &word = ctx->pieces_mask[chunk >> 6];
    const uint64_t bit = (uint64_t)1 << (chunk & 63);
    if (!(word & bit)) {
        word |= bit;
        ++ctx->pieces_seen;
    }
    if (ctx->pieces_seen == ctx->pieces_total)
        ctx->pieces_seen = ctx->pieces_total = 0; /* the step is complete */
    return RT_OK;
}

int rt_set_keep_rows(rt_ctx *ctx, const unsigned char *keep, int n)
{
    if (!ctx || (keep && (n < 1 || n > RT_MAX_SURFACES)))
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_keep_rows: bad argument");
    unsigned char want[RT_MAX_SURFACES];
    memset(want, 1, sizeof want);
    if (keep)
        for (int j = 0; j < n; ++j)
            want[j] = keep[j] ? 1 : 0;
    if (memcmp(want, ctx->keep, sizeof want)) {
        memcpy(ctx->keep, want, sizeof want);
        ctx->table_dirty = 1;
    }
    return RT_OK;
}

int rt_sync(rt_ctx *ctx)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_sync: NULL context");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_kernel_ms(rt_ctx *ctx, double *ms)
{
    if (!ctx || !ms)
        return rt_fail(ctx, RT_ERR_ARG, "rt_kernel_ms: NULL argument");
    if (!ctx->traced)
        return rt_fail(ctx, RT_ERR_STATE, "rt_kernel_ms: nothing traced yet");
    RT_HIP(ctx, hipEventSynchronize(ctx->k1));
    float f = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->k0, ctx->k1));
    *ms = f;
    return RT_OK;
}

int rt_event_record(rt_ctx *ctx, int slot)
{
    if (!ctx || slot < 0 || slot >= RT_NEVENTS)
        return rt_fail(ctx, RT_ERR_ARG, "rt_event_record: bad slot");
    RT_HIP(ctx, hipEventRecord(ctx->ev[slot], ctx->stream));
    return RT_OK;
}

int rt_event_elapsed(rt_ctx *ctx, int a, int b, double *ms)
{
    if (!ctx || !ms || a < 0 || a >= RT_NEVENTS || b < 0 || b >= RT_NEVENTS)
        return rt_fail(ctx, RT_ERR_ARG, "rt_event_elapsed: bad argument");
    RT_HIP(ctx, hipEventSynchronize(ctx->ev[b]));
    float f = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->ev[a], ctx->ev[b]));
    *ms = f;
    return RT_OK;
}

int rt_set_option(rt_ctx *ctx, const char *key, int value)
{
    if (!ctx || !key)
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_option: NULL argument");
    if (!strcmp(key, "alias_i")) {
        ctx->opt_alias = value ? 1 : 0;
        ctx->table_dirty = 1;
    } else if (!strcmp(key, "fuse_generate")) {
        ctx->opt_fuse = value ? 1 : 0;
    } else if (!strcmp(key, "regenerate")) {
        ctx->opt_regen = value ? 1 : 0;
    } else if (!strcmp(key, "fast_asphere") || !strcmp(key, "exact_asphere")) {
        const int fast = !strcmp(key, "fast_asphere") ? value != 0 : value == 0;
        if (fast != ctx->opt_fast)
            ctx->table_dirty = 1;
        ctx->opt_fast = fast;
    } else if (!strcmp(key, "range_shortcuts")) {
        if ((value != 0) != ctx->opt_range)
            ctx->table_dirty = 1;
        ctx->opt_range = value ? 1 : 0;
    } else if (!strcmp(key, "uniform_input")) {
        ctx->opt_uniform = value ? 1 : 0;
    } else if (!strcmp(key, "block_rays")) {
        /* takes effect with the next rt_reserve / rt_set_rays */
        if (value < 0)
            return rt_fail(ctx, RT_ERR_ARG, "block_rays: 0 (automatic) or a "
                                            "number of rays");
        ctx->opt_block = value;
    } else if (!strcmp(key, "pitch_rays")) {
        /* takes effect with the next rt_reserve: the row pitch is rounded up
         * to a multiple of this many rays (a multiple of 256; 0: the
         * library's choice) */
        if (value < 0 || value % 256)
            return rt_fail(ctx, RT_ERR_ARG, "pitch_rays: 0 or a multiple of "
                                            "256 rays");
        ctx->opt_pitch = value;
    } else if (!strcmp(key, "turn_points")) {
        /* pupil points per turn of a generated batch of several bundles
         * (rt_gen_wg): 0 automatic, -1 never, else that many (tests) */
        if (value < -1)
            return rt_fail(ctx, RT_ERR_ARG, "turn_points: -1, 0 or a number "
                                            "of pupil points");
        ctx->opt_turn = value;
    } else if (!strcmp(key, "consumers_one_pass")) {
        ctx->opt_onepass = value ? 1 : 0;
    } else if (!strcmp(key, "consumer_events")) {
        ctx->opt_cevents = value ? 1 : 0;
    } else if (!strcmp(key, "placement")) {
        /* takes effect with the next allocation of the result arrays */
        ctx->opt_place = value ? 1 : 0;
    } else if (!strcmp(key, "placement_orders")) {
        /* orders of a set's pieces along the range that are tried while the
         * store pattern is below the mark; -1: the library's choice */
        if (value < -1 || value > 64)
            return rt_fail(ctx, RT_ERR_ARG, "placement_orders: -1 .. 64");
        ctx->opt_place_orders = value;
    } else if (!strcmp(key, "placement_budget_ms")) {
        /* wall time after which an allocation stops CHOOSING memory (surplus
         * pieces, hops, further sets); 0: the default of 250 ms */
        if (value < 0)
            return rt_fail(ctx, RT_ERR_ARG, "placement_budget_ms: >= 0");
        ctx->opt_place_budget_ms = value ? (float)value : RT_PLACE_BUDGET_MS;
    } else if (!strcmp(key, "placement_good_gbps")) {
        /* takes effect with the next allocation (tests: a value no memory
         * reaches makes every allocation try all its ranges and sets) */
        if (value < 0)
            return rt_fail(ctx, RT_ERR_ARG, "placement_good_gbps: >= 0");
        ctx->opt_place_good = (float)value;
    } else if (!strcmp(key, "resident_lds")) {
        if (value < -1 || value > 65536)
            return rt_fail(ctx, RT_ERR_ARG,
                           "resident_lds must be -1 (auto) or 0..65536 bytes");
        ctx->opt_resident = value;
    } else if (!strcmp(key, "compact")) {
        if (value < 0 || value > 2)
            return rt_fail(ctx, RT_ERR_ARG, "compact must be 0, 1 or 2");
        ctx->opt_compact = value;
    } else if (!strcmp(key, "compact_every")) {
        if (value < 1 || value > 64)
            return rt_fail(ctx, RT_ERR_ARG, "compact_every must be in [1, 64]");
        ctx->opt_compact_every = value;
    } else {
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_option: unknown key '%s'", key);
    }
    return RT_OK;
}


int rt_download(rt_ctx *ctx, int which, int surf_lo, int surf_hi, double *dst)
{
    if (ctx)
        RT_ROWS_WHOLE(ctx, "rt_download");
    if (!ctx || !dst || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download: bad argument");
    if (!ctx->d_buf || surf_lo < 0 || surf_hi > ctx->buf_nsurf ||
        surf_lo >= surf_hi)
        return rt_fail(ctx, RT_ERR_STATE, "rt_download: rows [%d,%d) of %d",
                       surf_lo, surf_hi, ctx->buf_nsurf);
    const int nc = rt_ncomp(which);
    for (int j = surf_lo; j < surf_hi; ++j)
        if (!ctx->valid[j])
            return rt_fail(ctx, RT_ERR_STATE,
                           "rt_download: row %d holds no data (not kept by "
                           "rt_set_keep_rows, or not traced yet)", j);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rt_gen_flush(ctx);
    if (rc != RT_OK)
        return rc;
    if (which == RT_Y || which == RT_T) {
        rc = rt_rows_to_host(ctx, dst, rt_row(ctx, which, surf_lo),
                             (size_t)(surf_hi - surf_lo) * nc);
    } else { /* rows of I may live in U and rows of U in I: one by one */
        for (int j = surf_lo; j < surf_hi && rc == RT_OK; ++j)
            rc = rt_rows_to_host(ctx,
                                 dst + (size_t)(j - surf_lo) * nc * ctx->n,
                                 rt_row(ctx, which, j), nc);
    }
    if (rc != RT_OK)
        return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

/* x and y of ONE row: what the spot consumers of the host read
 * (`t.y[-1, :, :2]`, rayopt/geometric_trace.py:172, rayopt/analysis.py:237-283)
 * is two thirds of the row -- 160 instead of 240 MB over PCIe at 10^7 rays */
int rt_download_xy(rt_ctx *ctx, int which, int surf, double *dst)
{
    if (ctx)
        RT_ROWS_WHOLE(ctx, "rt_download_xy");
    if (!ctx || !dst || which < RT_Y || which > RT_I)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download_xy: bad argument");
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_download_xy: row %d of %d", surf,
                       ctx->buf_nsurf);
    if (!ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_download_xy: row %d holds no data (not kept by "
                       "rt_set_keep_rows, or not traced yet)", surf);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rt_gen_flush(ctx);
    if (rc != RT_OK)
        return rc;
    rc = rt_rows_to_host(ctx, dst, rt_row(ctx, which, surf), 2);
    if (rc != RT_OK)
        return rc;
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_download_ray(rt_ctx *ctx, int which, int64_t ray, double *dst)
{
    if (ctx)
        RT_ROWS_WHOLE(ctx, "rt_download_ray");
    if (!ctx || !dst || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download_ray: bad argument");
    if (!ctx->d_buf || ray < 0 || ray >= ctx->n)
        return rt_fail(ctx, RT_ERR_STATE, "rt_download_ray: ray %lld of %lld",
                       (long long)ray, (long long)ctx->n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    const int nc = rt_ncomp(which);
    for (int j = 0; j < ctx->buf_nsurf; ++j) {
        if (!ctx->valid[j]) { /* row not stored: NaN, like a dead ray */
            for (int c = 0; c < nc; ++c)
                dst[(size_t)j * nc + c] = __builtin_nan("");
            continue;
        }
        RT_HIP(ctx, hipMemcpy2DAsync(dst + (size_t)j * nc, sizeof(double),
                                     rt_row(ctx, which, j) +
                                         rt_block_col(ctx->bs, ctx->bts, ray),
                                     ctx->bs * sizeof(double), sizeof(double),
                                     nc, hipMemcpyDeviceToHost, ctx->stream));
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

/* rays ray0, ray0 + stride, ... of one row (nc components) -> out[nc][count] */
__global__ void rt_gather_rays_kernel(const double *__restrict__ row,
                                      int64_t bs, int64_t bts, int nc,
                                      int64_t ray0, int64_t stride,
                                      int64_t count, double *__restrict__ out)
{
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= count)
        return;
    const int64_t col = rt_block_col(bs, bts, ray0 + k * stride);
    for (int c = 0; c < nc; ++c)
        out[(int64_t)c * count + k] = row[(int64_t)c * bs + col];
}

int rt_download_rays(rt_ctx *ctx, int which, int64_t ray0, int64_t stride,
                     int64_t count, double *dst)
{
    if (ctx)
        RT_ROWS_WHOLE(ctx, "rt_download_rays");
    if (!ctx || !dst || which < RT_Y || which > RT_T || stride < 1 ||
        count < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download_rays: bad argument");
    if (!ctx->d_buf || ray0 < 0 || ray0 + (count - 1) * stride >= ctx->n)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_download_rays: rays %lld + k * %lld, k < %lld, of "
                       "%lld", (long long)ray0, (long long)stride,
                       (long long)count, (long long)ctx->n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = rt_gen_flush(ctx);
    if (rc != RT_OK)
        return rc;
    const int nc = rt_ncomp(which), L = ctx->buf_nsurf;
    const size_t bytes = (size_t)L * nc * (size_t)count * sizeof(double);
    rc = rt_need_scratch(ctx, bytes);
    if (rc != RT_OK)
        return rc;
    double *tmp = (double *)ctx->d_scratch;
    for (int j = 0; j < L; ++j) {
        double *out = tmp + (size_t)j * nc * (size_t)count;
        if (!ctx->valid[j]) { /* row not stored: NaN, like a dead ray */
            RT_HIP(ctx, hipMemsetAsync(out, 0xff, (size_t)nc * count *
                                                      sizeof(double),
                                       ctx->stream));
            continue;
        }
        hipLaunchKernelGGL(rt_gather_rays_kernel,
                           dim3((unsigned)((count + 255) / 256)), dim3(256), 0,
                           ctx->stream, rt_row(ctx, which, j), ctx->bs,
                           ctx->bts, nc, ray0, stride, count, out);
    }
    RT_HIP(ctx, hipGetLastError());
    return rt_d2h(ctx, dst, tmp, bytes);
}

int rt_device_ptr(rt_ctx *ctx, int which, int surf, void **out)
{
    if (ctx)
        RT_ROWS_WHOLE(ctx, "rt_device_ptr");
    if (!ctx || !out || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_device_ptr: bad argument");
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_device_ptr: no such row %d", surf);
    if (!ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE, "rt_device_ptr: row %d holds no data",
                       surf);
    int rc = rt_gen_flush(ctx);
    if (rc != RT_OK)
        return rc;
    if (surf == 0) {
        ctx->gen_live = 0; /* the caller may write through the pointer */
        ctx->uni_valid = 0;
    }
    *out = rt_row(ctx, which, surf);
    return RT_OK;
}

int rt_input_uniform(rt_ctx *ctx, int64_t *tiles7)
{
    if (!ctx || !tiles7)
        return rt_fail(ctx, RT_ERR_ARG, "rt_input_uniform: NULL argument");
    for (int c = 0; c < 7; ++c)
        tiles7[c] = 0;
    if (!ctx->d_buf || ctx->ld < 64)
        return RT_OK;
    tiles7[6] = ctx->ld / 64;
    if (!ctx->uni_valid || !ctx->opt_uniform)
        return RT_OK;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned *host = (unsigned *)malloc((size_t)tiles7[6] * sizeof(unsigned));
    if (!host)
        return rt_fail(ctx, RT_ERR_NOMEM, "rt_input_uniform: host allocation");
    hipError_t e = hipMemcpyAsync(host, ctx->d_uni,
                                  (size_t)tiles7[6] * sizeof(unsigned),
                                  hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        free(host);
        return rt_fail(ctx, RT_ERR_HIP, "rt_input_uniform: %s",
                       hipGetErrorString(e));
    }
    for (int64_t t = 0; t < tiles7[6]; ++t)
        for (int c = 0; c < 6; ++c)
            tiles7[c] += (host[t] >> c) & 1u;
    free(host);
    return RT_OK;
}

/* tiles of row 0 whose u_z a trace rebuilds from u_x, u_y instead of reading
 * it (the seed kernels found it bit for bit the completion of the other two;
 * tiles in which u_z is uniform anyway are not counted) */
int rt_input_completed(rt_ctx *ctx, int64_t *tiles)
{
    if (!ctx || !tiles)
        return rt_fail(ctx, RT_ERR_ARG, "rt_input_completed: NULL argument");
    *tiles = 0;
    if (!ctx->d_buf || ctx->ld < 64 || !ctx->uni_valid || !ctx->opt_uniform)
        return RT_OK;
    const int64_t nt = ctx->ld / 64;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned *host = (unsigned *)malloc((size_t)nt * sizeof(unsigned));
    if (!host)
        return rt_fail(ctx, RT_ERR_NOMEM, "rt_input_completed: host allocation");
    hipError_t e = hipMemcpyAsync(host, ctx->d_uni, (size_t)nt * sizeof(unsigned),
                                  hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        free(host);
        return rt_fail(ctx, RT_ERR_HIP, "rt_input_completed: %s",
                       hipGetErrorString(e));
    }
    for (int64_t t = 0; t < nt; ++t)
        *tiles += !(host[t] & 32u) &&
                  (host[t] & (RT_NOTE_UZ_A | RT_NOTE_UZ_B)) != 0;
    free(host);
    return RT_OK;
}

int rt_selftest_arith(rt_ctx *ctx, uint64_t seed, int64_t n, int span,
                      uint64_t mismatches[4])
{
    if (!ctx || !mismatches || n < 1 || span < 1 || span > 1000)
        return rt_fail(ctx, RT_ERR_ARG, "rt_selftest_arith: bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long *d = NULL;
    RT_HIP(ctx, hipMalloc((void **)&d, 4 * sizeof *d));
    hipError_t e = hipMemsetAsync(d, 0, 4 * sizeof *d, ctx->stream);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_selftest_kernel,
                           dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                           ctx->stream, (unsigned long long)seed,
                           (long long)n, span, d);
        e = hipGetLastError();
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess)
        e = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    if (e != hipSuccess)
        return rt_fail(ctx, RT_ERR_HIP, "rt_selftest_arith: %s",
                       hipGetErrorString(e));
    for (int k = 0; k < 4; ++k)
        mismatches[k] = h[k];
    return RT_OK;
}

int rt_placement(rt_ctx *ctx, int info[16], double ms[16])
{
    if (!ctx || !info || !ms)
        return rt_fail(ctx, RT_ERR_ARG, "rt_placement: NULL argument");
    const rt_place &p = ctx->place;
    memset(info, 0, 16 * sizeof(int));
    memset(ms, 0, 16 * sizeof(double));
    info[0] = p.base ? p.n : 0;
    info[1] = (int)(p.piece >> 20);
    info[2] = p.created;
    info[3] = p.nclass;
    for (int k = 0; k < 3; ++k)
        info[4 + k] = p.count[k];
    info[7] = p.fast;
    info[8] = p.ballast;
    info[9] = p.class_mix;
    info[10] = p.cut_short;
    info[11] = p.settled;
    info[12] = p.base ? p.picks : 0;
    info[13] = ctx->place_incoherent;
    info[14] = g_place_vm_failures;
    info[15] = p.base ? p.orders : 0;
    ms[0] = p.self_ms;
    ms[1] = p.cross_ms;
    ms[2] = p.store_gbps;
    ms[3] = p.search_ms;
    ms[4] = p.pieces_ms;
    ms[5] = p.ballast_ms;
    ms[6] = p.remap_ms;
    ms[7] = p.tune_ms;
    for (int k = 0; k < 5; ++k)
        ms[8 + k] = p.pick_gbps[k];
    ms[13] = p.slowest_create_ms;
    ms[14] = p.total_ms;
    return RT_OK;
}

int rt_scratch(rt_ctx *ctx, int64_t bytes, void **out)
{
    if (!ctx || !out || bytes < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_scratch: bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if ((size_t)bytes > ctx->user_bytes) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
        if (ctx->d_user)
            RT_HIP(ctx, hipFree(ctx->d_user));
        ctx->d_user = NULL;
        ctx->user_bytes = 0;
        hipError_t e = hipMalloc(&ctx->d_user, (size_t)bytes);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return rt_fail(ctx, RT_ERR_NOMEM, "rt_scratch: hipMalloc(%lld): %s",
                           (long long)bytes, hipGetErrorString(e));
        }
        ctx->user_bytes = (size_t)bytes;
    }
    *out = ctx->d_user;
    return RT_OK;
}

int rt_copy_to_host(rt_ctx *ctx, void *dst, const void *d_src, int64_t bytes)
{
    if (!ctx || !dst || !d_src || bytes < 0)
        return rt_fail(ctx, RT_ERR_ARG, "rt_copy_to_host: bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    return rt_d2h(ctx, dst, d_src, (size_t)bytes);
}

} /* extern "C" */
