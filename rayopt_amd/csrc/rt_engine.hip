/*
 * rt_engine.hip -- gfx950 kernels + C ABI (include/rt_mi355.h) of the
 * sequential geometric ray-trace engine.
 *
 * Kernel design (MI355X first):
 *  - one lane owns one ray (R = 1: 8-byte global accesses, 512 B per wave
 *    instruction; 2 or 4 adjacent rays per lane are template variants) and
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
 *    is decided per wavefront with a 64-bit ballot (rt_math.h); an opt-in fast
 *    arithmetic (FMA, rcp/rsq, one reciprocal per iterate) takes it off the
 *    FP64-issue wall;
 *  - a wavefront whose rays are all dead stores NaN rows without evaluating
 *    the element; an opt-in kernel compacts the survivors of a workgroup into
 *    fewer wavefronts (ballots + LDS) for traces that keep few rows;
 *  - FP64 VALU only: the path is elementwise, there is no contraction to put
 *    on MFMA.  Bound: HBM write bandwidth (56-80 B / ray-surface op).
 *
 * No CPU fallback lives here: every entry point either runs on the GPU or
 * returns an error.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>

#include "rt_math.h"

#include "rt_kernels.h"

/* ------------------------------------------------------------------ */
/* context                                                            */
/* ------------------------------------------------------------------ */

#define RT_NEVENTS 8
#define RT_MAX_GROUPS 65535 /* surface tables per launch (wavelengths, or
                               variants of a system: tolerancing runs) */

struct rt_rccl_api {
    void *lib;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *);
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int);
    ncclResult_t (*CommDestroy)(ncclComm_t);
    ncclResult_t (*GroupStart)(void);
    ncclResult_t (*GroupEnd)(void);
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t,
                         hipStream_t);
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t,
                         hipStream_t);
    const char *(*GetErrorString)(ncclResult_t);
};

struct rt_ctx {
    int device;
    hipStream_t stream;      /* trace + copies */
    hipStream_t comm_stream; /* RCCL gather */
    hipEvent_t k0, k1;       /* around the last trace kernel */
    hipEvent_t ev[RT_NEVENTS];
    int traced;

    rt_surface *d_surf;  /* = d_tab[tab_cur]: the table kernels read */
    rt_surface *d_tab[2]; /* double buffered: a changed table is sent while a
                             kernel in flight still reads the previous one */
    hipEvent_t tab_used[2]; /* last DMA into / kernel reading buffer k */
    int tab_cur;
    int nsurf;
    rt_surface *h_surf;                 /* [ngroups][nsurf] as given */
    size_t tab_cap;                     /* entries h_surf/h_stage/d_surf hold */
    int ngroups;                        /* surface tables (wavelengths) */
    rt_surface *h_stage;                /* = h_pinned[tab_cur] */
    rt_surface *h_pinned[2];            /* pinned: flags finalised */
    int table_dirty;
    int table_start;
    unsigned char keep[RT_MAX_SURFACES];  /* rows propagate() stores */
    unsigned char valid[RT_MAX_SURFACES]; /* rows that hold data */

    double *d_buf; /* Y | U | I | T */
    size_t cap_doubles;
    int64_t n, ld;
    int buf_nsurf; /* L the buffer is laid out for */

    void *d_scratch;
    size_t scratch_bytes;
    void *d_user; /* rt_scratch */
    size_t user_bytes;
    void *h_pin[2]; /* pinned staging for large pageable copies */
    hipEvent_t pin_done[2];
    int pin_busy[2]; /* a DMA recorded in pin_done[k] may still use h_pin[k] */
    char *h_aim; /* pinned: rt_aim_pupil's tables | seeds out, z | a | status in */
    size_t h_aim_bytes;
    double *d_w;  /* ray weights, NULL = uniform 1/n */
    size_t w_cap;
    int64_t w_n;  /* rays the weights were given for (must equal n) */
    double *d_partials; /* RT_RED_BLOCKS x 8 doubles + 16 reduced values */
    double *d_group;    /* rt_spot_stats: stats | partials */
    size_t group_cap;   /* doubles */
    rt_opd_ref *d_opd_ref;

    /* kernel variant */
    int opt_r, opt_nt, opt_xcd, opt_block, opt_alias;
    int opt_lds; /* bytes of unused dynamic LDS per workgroup (occupancy) */
    int opt_fuse; /* build generated rays inside the first trace */
    int opt_fast; /* aspheric elements on the fast arithmetic (RT_F_FAST) */
    int opt_tile; /* measurement only: tile-major result layout, rays/tile */
    int opt_uniform_fix; /* measurement only: input components read as if
                            wave-uniform (6-bit mask) */
    int opt_gate_log2, opt_gate_window; /* measurement only: read windows */
    int opt_probe_store; /* rt_probe pattern modes: 0 plain 1 nt 2 sc1 3 sc0sc1 */
    void *d_probe_in; /* rt_probe modes 13/14: input rows of their own */
    size_t probe_in_bytes;
    int probe_in_uc;
    int opt_compact; /* 0 never, 1 when rows are dropped, 2 always */
    int opt_compact_every; /* survivors are counted at every k-th element */
    int last_compact; /* the last trace ran the compacting kernel */

    /* rt_generate_rays: field frames | pupil points, and whether row 0 is
     * still to be built from them */
    void *d_gen;
    size_t gen_bytes, gen_fpad;
    int gen_pending, gen_nf;
    int gen_live; /* row 0 still holds exactly what d_gen describes: a trace
                     from element 1 may rebuild the rays instead of reading
                     them */
    int opt_regen;
    int64_t gen_np, gen_n;
    rt_surface gen_s0;
    /* per row of I: 0 = materialised, 1 = identical to U[j-1], 2 = to U[j] */
    unsigned char i_alias[RT_MAX_SURFACES];
    /* per row of U: 1 = identical to I[j] (RT_F_SKIP_U), not materialised */
    unsigned char u_alias[RT_MAX_SURFACES];
    int table_clip; /* clip the device table was finalised for */

    /* multi GPU */
    ncclComm_t comm;
    int nranks, rank;
    double *d_stage[2];
    size_t stage_bytes;
    hipEvent_t staged[2], gathered[2];
    int gather_pending[2];
    int parity;

    char err[512];
};

static char g_err[512] = "";
static rt_rccl_api g_rccl = {};

static int rt_fail(rt_ctx *ctx, int code, const char *fmt, ...)
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

#define RT_HIP(ctx, call)                                                     \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess)                                                 \
            return rt_fail(ctx, RT_ERR_HIP, "%s: %s (%s:%d)", #call,          \
                           hipGetErrorString(e_), __FILE__, __LINE__);        \
    } while (0)

#define RT_NCCL(ctx, call)                                                    \
    do {                                                                      \
        ncclResult_t r_ = (call);                                             \
        if (r_ != ncclSuccess)                                                \
            return rt_fail(ctx, RT_ERR_RCCL, "%s: %s (%s:%d)", #call,         \
                           g_rccl.GetErrorString(r_), __FILE__, __LINE__);    \
    } while (0)

static inline double *rt_arr(const rt_ctx *c, int which)
{
    /* Y,U,I are [L][3][ld]; T is [L][ld] */
    const size_t plane = (size_t)c->buf_nsurf * 3 * (size_t)c->ld;
    return c->d_buf + (size_t)which * plane;
}

static inline int rt_ncomp(int which) { return which == RT_T ? 1 : 3; }

/* addressing of the result arrays as the kernels see it (rt_kernels.h) */
static inline rt_lay rt_layout(const rt_ctx *c)
{
    rt_lay a;
    if (!c->opt_tile) {
        a.Y = rt_arr(c, RT_Y);
        a.U = rt_arr(c, RT_U);
        a.I = rt_arr(c, RT_I);
        a.T = rt_arr(c, RT_T);
        a.cs = c->ld;
        a.ss = 3 * c->ld;
        a.ssT = c->ld;
        a.tshift = 8;
        a.ts = 256;
    } else { /* [tile][L][10][TR] */
        const int64_t tr = c->opt_tile;
        a.Y = c->d_buf;
        a.U = c->d_buf + 3 * tr;
        a.I = c->d_buf + 6 * tr;
        a.T = c->d_buf + 9 * tr;
        a.cs = tr;
        a.ss = a.ssT = 10 * tr;
        a.ts = (int64_t)c->buf_nsurf * 10 * tr;
        a.tshift = __builtin_ctzll((unsigned long long)tr);
    }
    return a;
}

/* everything that reads rows back assumes the documented SoA layout */
static int rt_soa_only(rt_ctx *c, const char *who)
{
    if (c && c->opt_tile)
        return rt_fail(c, RT_ERR_STATE,
                       "%s: the tile_rays layout is a measurement option; "
                       "results can only be read back in the SoA layout", who);
    return RT_OK;
}

/* device address of one surface row, resolving the I -> U aliasing */
static inline double *rt_row(const rt_ctx *c, int which, int surf)
{
    if (which == RT_I && c->i_alias[surf])
        return rt_row(c, RT_U, c->i_alias[surf] == 1 ? surf - 1 : surf);
    if (which == RT_U && c->u_alias[surf])
        return rt_row(c, RT_I, surf); /* surf >= 1, and I[surf] never points
                                         back at U[surf] there */
    return rt_arr(c, which) + (size_t)surf * rt_ncomp(which) * c->ld;
}

/* give row `surf` of U or I its own copy of the data it is served from (a
 * kernel is about to read it at its natural address, or the row it is served
 * from is about to be overwritten) */
static int rt_detach(rt_ctx *c, int which, int surf)
{
    unsigned char *alias = which == RT_U ? c->u_alias : c->i_alias;
    if ((which != RT_U && which != RT_I) || !alias[surf])
        return RT_OK;
    const double *src = rt_row(c, which, surf);
    alias[surf] = 0;
    double *dst = rt_row(c, which, surf);
    RT_HIP(c, hipMemcpyAsync(dst, src, (size_t)3 * c->ld * sizeof(double),
                             hipMemcpyDeviceToDevice, c->stream));
    return RT_OK;
}

/* row 0 of a generated batch, if no trace has built it yet */
static int rt_gen_flush(rt_ctx *c)
{
    if (!c || !c->gen_pending)
        return RT_OK;
    c->gen_pending = 0;
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

/* the compacting variant pays (one barrier per element) only where dead rays
 * are wasted FP64 issue, i.e. where rows are traced but not stored */
static bool rt_use_compact(const rt_ctx *c, int start, int stop)
{
    if (!c->opt_compact || c->opt_r != 1 || c->opt_nt || c->opt_xcd ||
        c->opt_tile)
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

template <int R, bool NT, bool XCD>
static void rt_launch(rt_ctx *c, int start, int stop, int clip)
{
    const int block = c->opt_block;
    const int64_t per_block = (int64_t)block * R;
    const int64_t nblocks = (c->ld + per_block - 1) / per_block;
    const int64_t grid = XCD ? (nblocks + 7) / 8 * 8 : nblocks;
    hipLaunchKernelGGL((rt_trace_kernel<R, NT, XCD>), dim3((unsigned)grid),
                       dim3(block), (size_t)c->opt_lds, c->stream, c->d_surf,
                       start, stop, clip, rt_layout(c), c->ld, nblocks,
                       c->ngroups > 1 ? c->n / c->ngroups : (int64_t)0,
                       c->nsurf, (const unsigned *)NULL,
                       (unsigned)(start == 1 ? c->opt_uniform_fix : 0),
                       c->opt_gate_log2 ? (1u << c->opt_gate_log2) - 1u : 0u,
                       (unsigned)c->opt_gate_window);
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
    c->opt_r = 1;
    c->opt_nt = 0;
    c->opt_xcd = 0;
    c->opt_block = 256;
    c->opt_alias = 1;
    c->opt_fuse = 1;
    c->opt_regen = 1;
    c->opt_compact_every = 4; /* measured best, profiles/r02_probes */
#define RT_HIP_C(call)                                                        \
    do {                                                                      \
        hipError_t e_ = (call);                                               \
        if (e_ != hipSuccess) {                                               \
            rt_fail(NULL, RT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_)); \
            free(c);                                                          \
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
    for (int i = 0; i < 2; ++i) {
        RT_HIP_C(hipEventCreateWithFlags(&c->staged[i], hipEventDisableTiming));
        RT_HIP_C(
            hipEventCreateWithFlags(&c->gathered[i], hipEventDisableTiming));
    }
    c->ngroups = 1;
    c->tab_cap = (size_t)4 * RT_MAX_SURFACES; /* grows in rt_upload_system */
    c->h_surf = (rt_surface *)calloc(c->tab_cap, sizeof(rt_surface));
    if (!c->h_surf) {
        free(c);
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

int rt_comm_destroy(rt_ctx *ctx);

int rt_destroy(rt_ctx *ctx)
{
    if (!ctx)
        return RT_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->comm_stream);
    rt_comm_destroy(ctx);
    if (ctx->d_buf)
        (void)hipFree(ctx->d_buf);
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
    if (ctx->d_group)
        (void)hipFree(ctx->d_group);
    if (ctx->d_gen)
        (void)hipFree(ctx->d_gen);
    if (ctx->d_probe_in)
        (void)hipFree(ctx->d_probe_in);
    if (ctx->d_opd_ref)
        (void)hipFree(ctx->d_opd_ref);
    for (int k = 0; k < 2; ++k) {
        if (ctx->d_tab[k])
            (void)hipFree(ctx->d_tab[k]);
        if (ctx->h_pinned[k])
            (void)hipHostFree(ctx->h_pinned[k]);
        (void)hipEventDestroy(ctx->tab_used[k]);
    }
    free(ctx->h_surf);
    for (int i = 0; i < 2; ++i) {
        if (ctx->d_stage[i])
            (void)hipFree(ctx->d_stage[i]);
        (void)hipEventDestroy(ctx->staged[i]);
        (void)hipEventDestroy(ctx->gathered[i]);
    }
    for (int i = 0; i < RT_NEVENTS; ++i)
        (void)hipEventDestroy(ctx->ev[i]);
    (void)hipEventDestroy(ctx->k0);
    (void)hipEventDestroy(ctx->k1);
    (void)hipStreamDestroy(ctx->stream);
    (void)hipStreamDestroy(ctx->comm_stream);
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
    ctx->table_dirty = 1; /* finalised and sent by the next rt_trace */
    ctx->nsurf = nsurf;
    ctx->ngroups = ngroups;
    return RT_OK;
}

int rt_upload_system(rt_ctx *ctx, const rt_surface *surf, int nsurf)
{
    return rt_upload_system_groups(ctx, surf, nsurf, 1);
}

int rt_reserve(rt_ctx *ctx, int64_t nrays)
{
    if (!ctx || nrays < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_reserve: bad argument");
    if (ctx->nsurf < 2)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_reserve: rt_upload_system must come first");
    const int64_t quantum = ctx->opt_tile ? ctx->opt_tile : 64;
    const int64_t ld = (nrays + quantum - 1) / quantum * quantum;
    if (ld == ctx->ld && ctx->buf_nsurf == ctx->nsurf && ctx->d_buf) {
        if (nrays != ctx->n)
            ctx->gen_live = 0;
        ctx->n = nrays;
        return RT_OK;
    }
    ctx->gen_live = 0;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t need = (size_t)ctx->nsurf * 10 * (size_t)ld;
    if (need > ctx->cap_doubles) {
        if (ctx->d_buf)
            RT_HIP(ctx, hipFree(ctx->d_buf));
        ctx->d_buf = NULL;
        ctx->cap_doubles = 0;
        hipError_t e = hipMalloc((void **)&ctx->d_buf, need * sizeof(double));
        if (e != hipSuccess) {
            (void)hipGetLastError();
            return rt_fail(ctx, RT_ERR_NOMEM,
                           "rt_reserve: hipMalloc of %.3f GB failed: %s",
                           need * 8e-9, hipGetErrorString(e));
        }
        ctx->cap_doubles = need;
    }
    ctx->n = nrays;
    ctx->ld = ld;
    ctx->buf_nsurf = ctx->nsurf;
    ctx->traced = 0;
    memset(ctx->i_alias, 0, sizeof ctx->i_alias);
    memset(ctx->u_alias, 0, sizeof ctx->u_alias);
    memset(ctx->valid, 0, sizeof ctx->valid);
    return RT_OK;
}

int64_t rt_nrays(const rt_ctx *ctx) { return ctx ? ctx->n : 0; }
int64_t rt_ld(const rt_ctx *ctx) { return ctx ? ctx->ld : 0; }
int rt_nsurf(const rt_ctx *ctx) { return ctx ? ctx->nsurf : 0; }

static int rt_need_scratch(rt_ctx *ctx, size_t bytes)
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

static int rt_seed(rt_ctx *ctx, const double *d_y, const double *d_u,
                   int64_t n, int layout, int64_t period)
{
    const int block = 256;
    const unsigned grid = (unsigned)((ctx->ld + block - 1) / block);
    if (layout == RT_LAYOUT_AOS)
        hipLaunchKernelGGL(rt_seed_aos_kernel, dim3(grid), dim3(block), 0,
                           ctx->stream, d_y, d_u, n, rt_layout(ctx), ctx->ld,
                           !ctx->opt_alias, period);
    else
        hipLaunchKernelGGL(rt_seed_soa_kernel, dim3(grid), dim3(block), 0,
                           ctx->stream, d_y, d_u, n, rt_layout(ctx), ctx->ld,
                           !ctx->opt_alias, period);
    RT_HIP(ctx, hipGetLastError());
    ctx->i_alias[0] = ctx->opt_alias ? 2 : 0; /* i[0] = u[0] (:67) */
    ctx->u_alias[0] = 0;
    ctx->valid[0] = 1;
    ctx->gen_pending = 0; /* these rays replace a generated batch */
    ctx->gen_live = 0;
    return RT_OK;
}

#define RT_PIN_CHUNK ((size_t)32 << 20)

/*
 * memcpy between pageable memory and the pinned staging buffers on a few
 * threads: one core copies ~30 GB/s, the DMA engine moves ~55 GB/s over
 * PCIe 5 x16, so the single-threaded staging copy was the slower half of the
 * pipeline (RT_COPY_THREADS overrides the default of 4; 1 = plain memcpy).
 */
static int rt_copy_threads(void)
{
    static int n = 0;
    if (!n) {
        const char *e = getenv("RT_COPY_THREADS");
        n = e ? atoi(e) : 4;
        n = n < 1 ? 1 : (n > 16 ? 16 : n);
    }
    return n;
}

static void rt_memcpy_mt(void *dst, const void *src, size_t len)
{
    const int nt = rt_copy_threads();
    if (nt == 1 || len < ((size_t)4 << 20)) {
        memcpy(dst, src, len);
        return;
    }
    const size_t part = (len / nt + 4095) & ~(size_t)4095;
    std::thread workers[16];
    int started = 0;
    for (int t = 1; t < nt; ++t) {
        const size_t off = (size_t)t * part;
        if (off >= len)
            break;
        const size_t n = len - off < part ? len - off : part;
        workers[started++] = std::thread(
            [=] { memcpy((char *)dst + off, (const char *)src + off, n); });
    }
    memcpy(dst, src, part < len ? part : len);
    for (int t = 0; t < started; ++t)
        workers[t].join();
}

/*
 * Host -> device copy of a pageable buffer through two pinned staging
 * buffers: the CPU fills one while the DMA engine drains the other.  A plain
 * hipMemcpyAsync from pageable memory is staged by the runtime in small
 * pieces and reaches ~5 GB/s; this path is bound by the host memcpy.
 */
static int rt_h2d(rt_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes < RT_PIN_CHUNK / 8) {
        RT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice,
                                   ctx->stream));
        return RT_OK;
    }
    for (int i = 0; i < 2; ++i)
        if (!ctx->h_pin[i]) {
            RT_HIP(ctx, hipHostMalloc(&ctx->h_pin[i], RT_PIN_CHUNK));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_done[i],
                                                hipEventDisableTiming));
        }
    int k = 0;
    for (size_t off = 0; off < bytes; off += RT_PIN_CHUNK, k ^= 1) {
        const size_t len = bytes - off < RT_PIN_CHUNK ? bytes - off
                                                      : RT_PIN_CHUNK;
        if (ctx->pin_busy[k]) /* this call's or an earlier call's DMA */
            RT_HIP(ctx, hipEventSynchronize(ctx->pin_done[k]));
        rt_memcpy_mt(ctx->h_pin[k], (const char *)src + off, len);
        RT_HIP(ctx, hipMemcpyAsync((char *)dst + off, ctx->h_pin[k], len,
                                   hipMemcpyHostToDevice, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->pin_done[k], ctx->stream));
        ctx->pin_busy[k] = 1;
    }
    return RT_OK;
}

/* device -> pageable host, same double-buffered staging; synchronous */
static int rt_d2h(rt_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes < RT_PIN_CHUNK / 8) {
        RT_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost,
                                   ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return RT_OK;
    }
    for (int i = 0; i < 2; ++i)
        if (!ctx->h_pin[i]) {
            RT_HIP(ctx, hipHostMalloc(&ctx->h_pin[i], RT_PIN_CHUNK));
            RT_HIP(ctx, hipEventCreateWithFlags(&ctx->pin_done[i],
                                                hipEventDisableTiming));
        }
    const size_t nchunk = (bytes + RT_PIN_CHUNK - 1) / RT_PIN_CHUNK;
    for (size_t i = 0; i <= nchunk; ++i) {
        if (i < nchunk) { /* start the DMA of chunk i */
            const size_t off = i * RT_PIN_CHUNK;
            const size_t len = bytes - off < RT_PIN_CHUNK ? bytes - off
                                                          : RT_PIN_CHUNK;
            if (ctx->pin_busy[i & 1]) /* an upload may still read it */
                RT_HIP(ctx, hipEventSynchronize(ctx->pin_done[i & 1]));
            RT_HIP(ctx, hipMemcpyAsync(ctx->h_pin[i & 1],
                                       (const char *)src + off, len,
                                       hipMemcpyDeviceToHost, ctx->stream));
            RT_HIP(ctx, hipEventRecord(ctx->pin_done[i & 1], ctx->stream));
        }
        if (i > 0) { /* drain chunk i-1 while chunk i is in flight */
            const size_t off = (i - 1) * RT_PIN_CHUNK;
            const size_t len = bytes - off < RT_PIN_CHUNK ? bytes - off
                                                          : RT_PIN_CHUNK;
            RT_HIP(ctx, hipEventSynchronize(ctx->pin_done[(i - 1) & 1]));
            rt_memcpy_mt((char *)dst + off, ctx->h_pin[(i - 1) & 1], len);
            ctx->pin_busy[(i - 1) & 1] = 0;
        }
    }
    return RT_OK;
}

/* nrow rows of ld doubles -> compact rows of n doubles on the host */
static int rt_rows_to_host(rt_ctx *ctx, double *dst, const double *src,
                           size_t nrow)
{
    const size_t rb = (size_t)ctx->n * sizeof(double);
    if (ctx->ld == ctx->n) /* no padding: one contiguous block */
        return rt_d2h(ctx, dst, src, rb * nrow);
    if (rb >= RT_PIN_CHUNK / 8) {
        for (size_t r = 0; r < nrow; ++r) {
            int rc = rt_d2h(ctx, dst + r * ctx->n, src + r * ctx->ld, rb);
            if (rc != RT_OK)
                return rc;
        }
        return RT_OK;
    }
    RT_HIP(ctx, hipMemcpy2DAsync(dst, rb, src, ctx->ld * sizeof(double), rb,
                                 nrow, hipMemcpyDeviceToHost, ctx->stream));
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
    return RT_OK;
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

int rt_sizeof_aim_seed(void) { return (int)sizeof(rt_aim_seed); }
int rt_sizeof_aim_args(void) { return (int)sizeof(rt_aim_args); }

int rt_aim_pupil(rt_ctx *ctx, const rt_aim_seed *seeds, int nfields,
                 const rt_aim_args *args, double *z, double *a,
                 int32_t *status)
{
    if (!ctx || !seeds || !args || !z || !a || !status || nfields < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_aim_pupil: bad argument");
    if (ctx->nsurf < 3)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_aim_pupil: rt_upload_system must come first");
    if (args->stop < 1 || args->stop > ctx->nsurf - 2 || args->maxiter < 1)
        return rt_fail(ctx, RT_ERR_ARG, "rt_aim_pupil: stop %d of %d elements",
                       args->stop, ctx->nsurf);
    for (int f = 0; f < nfields; ++f)
        if (seeds[f].group < 0 || seeds[f].group >= ctx->ngroups)
            return rt_fail(ctx, RT_ERR_ARG,
                           "rt_aim_pupil: field %d names table %d of %d", f,
                           seeds[f].group, ctx->ngroups);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    /* scratch: tables | seeds | z | a | status, each 256-byte aligned; the
     * same layout in one pinned host buffer, so that a call costs one copy
     * in, the kernel, one copy out */
    const size_t ntab = (size_t)ctx->nsurf * ctx->ngroups;
    const size_t tb = (sizeof(rt_surface) * ntab + 255) / 256 * 256;
    const size_t sb = (sizeof(rt_aim_seed) * nfields + 255) / 256 * 256;
    const size_t zb = (sizeof(double) * nfields + 255) / 256 * 256;
    const size_t ab = (sizeof(double) * 4 * nfields + 255) / 256 * 256;
    const size_t cb = (sizeof(int32_t) * nfields + 255) / 256 * 256;
    const size_t all = tb + sb + zb + ab + cb;
    int rc = rt_need_scratch(ctx, all);
    if (rc != RT_OK)
        return rc;
    if (all > ctx->h_aim_bytes) {
        if (ctx->h_aim)
            (void)hipHostFree(ctx->h_aim);
        ctx->h_aim = NULL;
        ctx->h_aim_bytes = 0;
        const size_t want = all + all / 2;
        RT_HIP(ctx, hipHostMalloc((void **)&ctx->h_aim, want));
        ctx->h_aim_bytes = want;
    }
    char *base = (char *)ctx->d_scratch, *host = ctx->h_aim;
    rt_surface *d_tab = (rt_surface *)base;
    rt_aim_seed *d_seeds = (rt_aim_seed *)(base + tb);
    double *d_z = (double *)(base + tb + sb);
    double *d_a = (double *)(base + tb + sb + zb);
    int32_t *d_status = (int32_t *)(base + tb + sb + zb + ab);
    memcpy(host, ctx->h_surf, sizeof(rt_surface) * ntab);
    memcpy(host + tb, seeds, sizeof(rt_aim_seed) * nfields);
    RT_HIP(ctx, hipMemcpyAsync(base, host, tb + sb, hipMemcpyHostToDevice,
                               ctx->stream));
    /* one wavefront per field while they are all resident at once, 16
     * fields per wavefront beyond that (rt_kernels.h) */
    if (nfields <= 32768)
        hipLaunchKernelGGL(rt_aim_kernel<true>, dim3((unsigned)nfields),
                           dim3(4), 0, ctx->stream, d_tab, ctx->nsurf, d_seeds,
                           nfields, *args, d_z, d_a, d_status);
    else
        hipLaunchKernelGGL(rt_aim_kernel<false>,
                           dim3((unsigned)(((int64_t)nfields * 4 + 63) / 64)),
                           dim3(64), 0, ctx->stream, d_tab, ctx->nsurf,
                           d_seeds, nfields, *args, d_z, d_a, d_status);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(host + tb + sb, base + tb + sb, zb + ab + cb,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(z, host + tb + sb, sizeof(double) * nfields);
    memcpy(a, host + tb + sb + zb, sizeof(double) * 4 * nfields);
    memcpy(status, host + tb + sb + zb + ab, sizeof(int32_t) * nfields);
    return RT_OK;
}

int rt_upload_row(rt_ctx *ctx, int which, int surf, const double *src_soa)
{
    if (!ctx || !src_soa || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_upload_row: bad argument");
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_upload_row: no such row %d",
                       surf);
    if (rt_soa_only(ctx, "rt_upload_row") != RT_OK)
        return RT_ERR_STATE;
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
    if (surf == 0 && which != RT_I && which != RT_T)
        ctx->gen_live = 0; /* the launch rays are the caller's from here on */
    if (which == RT_I)
        ctx->i_alias[surf] = 0; /* now holds its own data */
    if (which == RT_U)
        ctx->u_alias[surf] = 0;
    ctx->valid[surf] = 1;
    double *dst = rt_row(ctx, which, surf);
    RT_HIP(ctx, hipMemcpy2DAsync(dst, ctx->ld * sizeof(double), src_soa,
                                 ctx->n * sizeof(double),
                                 ctx->n * sizeof(double), nc,
                                 hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_trace(rt_ctx *ctx, int start, int stop, int clip)
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
    if (ctx->ngroups > 1 && (ctx->n % ctx->ngroups ||
                             (ctx->n / ctx->ngroups) % (64 * ctx->opt_r)))
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_trace: %lld rays do not split into %d groups of a "
                       "multiple of %d rays", (long long)ctx->n, ctx->ngroups,
                       64 * ctx->opt_r);
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
        for (int jj = 0; jj < ntab; ++jj) {
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
            ctx->h_stage[jj].flags = f;
        }
        RT_HIP(ctx, hipMemcpyAsync(ctx->d_surf, ctx->h_stage,
                                   sizeof(rt_surface) * ntab,
                                   hipMemcpyHostToDevice, ctx->stream));
        ctx->table_dirty = 0;
        ctx->table_start = start;
        ctx->table_clip = clip != 0;
    } else if (ctx->table_start != start) {
        ctx->table_dirty = 1; /* alias decisions depend on start */
        return rt_trace(ctx, start, stop, clip);
    }
    /* a generated batch that no one has looked at yet is built inside this
     * launch (default kernel variant, from the first element on) */
    const bool gen_kernel = start == 1 && start < stop && ctx->opt_r == 1 &&
                            !ctx->opt_nt && !ctx->opt_xcd &&
                            !rt_use_compact(ctx, start, stop);
    const bool fused = ctx->gen_pending && gen_kernel;
    /* a later trace of the same generated batch builds the rays again in
     * registers (same frames, same arithmetic: the values row 0 holds)
     * rather than read 48 B per ray among the saturated stores */
    const bool regen = !ctx->gen_pending && ctx->gen_live && ctx->opt_regen &&
                       ctx->opt_fuse && gen_kernel && ctx->valid[0];
    if (!fused) {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    ctx->last_compact = 0;
    if (fused || regen) {
        const int block = ctx->opt_block;
        ctx->gen_pending = 0;
        hipLaunchKernelGGL(rt_trace_gen_kernel,
                           dim3((unsigned)((ctx->ld + block - 1) / block)),
                           dim3(block), (size_t)ctx->opt_lds, ctx->stream,
                           ctx->d_surf, stop, clip, rt_layout(ctx), ctx->ld,
                           ctx->ngroups > 1 ? ctx->n / ctx->ngroups
                                            : (int64_t)0,
                           ctx->nsurf, (const rt_field *)ctx->d_gen,
                           (const double *)((char *)ctx->d_gen + ctx->gen_fpad),
                           ctx->gen_np, ctx->gen_n, ctx->gen_s0,
                           !ctx->opt_alias, fused ? 1 : 0);
        RT_HIP(ctx, hipGetLastError());
    } else if (start < stop && rt_use_compact(ctx, start, stop)) {
        const unsigned grid = (unsigned)((ctx->ld + RT_CB - 1) / RT_CB);
        hipLaunchKernelGGL(rt_trace_compact_kernel, dim3(grid), dim3(RT_CB),
                           0, ctx->stream, ctx->d_surf, start, stop, clip,
                           rt_layout(ctx), ctx->ld,
                           ctx->ngroups > 1 ? ctx->n / ctx->ngroups
                                            : (int64_t)0,
                           ctx->nsurf, ctx->opt_compact_every);
        RT_HIP(ctx, hipGetLastError());
        ctx->last_compact = 1;
    } else if (start < stop) {
        const int key = ctx->opt_r * 4 + ctx->opt_nt * 2 + ctx->opt_xcd;
        switch (key) {
#define RT_CASE(R, NT, X)                                                     \
    case (R) * 4 + (NT) * 2 + (X):                                            \
        rt_launch<R, NT, X>(ctx, start, stop, clip);                          \
        break;
            RT_CASE(1, 0, 0) RT_CASE(1, 0, 1) RT_CASE(1, 1, 0) RT_CASE(1, 1, 1)
            RT_CASE(2, 0, 0) RT_CASE(2, 0, 1) RT_CASE(2, 1, 0) RT_CASE(2, 1, 1)
            RT_CASE(4, 0, 0) RT_CASE(4, 0, 1) RT_CASE(4, 1, 0) RT_CASE(4, 1, 1)
#undef RT_CASE
        default:
            return rt_fail(ctx, RT_ERR_STATE, "rt_trace: bad variant %d", key);
        }
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
    if (!strcmp(key, "rays_per_thread")) {
        if (value != 1 && value != 2 && value != 4)
            return rt_fail(ctx, RT_ERR_ARG, "rays_per_thread must be 1, 2, 4");
        ctx->opt_r = value;
    } else if (!strcmp(key, "nontemporal")) {
        ctx->opt_nt = value ? 1 : 0;
    } else if (!strcmp(key, "xcd_remap")) {
        ctx->opt_xcd = value ? 1 : 0;
    } else if (!strcmp(key, "alias_i")) {
        ctx->opt_alias = value ? 1 : 0;
        ctx->table_dirty = 1;
    } else if (!strcmp(key, "fuse_generate")) {
        ctx->opt_fuse = value ? 1 : 0;
    } else if (!strcmp(key, "regenerate")) {
        ctx->opt_regen = value ? 1 : 0;
    } else if (!strcmp(key, "fast_asphere")) {
        if ((value != 0) != ctx->opt_fast)
            ctx->table_dirty = 1;
        ctx->opt_fast = value ? 1 : 0;
    } else if (!strcmp(key, "compact")) {
        if (value < 0 || value > 2)
            return rt_fail(ctx, RT_ERR_ARG, "compact must be 0, 1 or 2");
        ctx->opt_compact = value;
    } else if (!strcmp(key, "compact_every")) {
        if (value < 1 || value > 64)
            return rt_fail(ctx, RT_ERR_ARG, "compact_every must be in [1, 64]");
        ctx->opt_compact_every = value;
    } else if (!strcmp(key, "gate_log2")) {
        if (value < 0 || value > 20)
            return rt_fail(ctx, RT_ERR_ARG, "gate_log2 must be in [0, 20]");
        ctx->opt_gate_log2 = value;
    } else if (!strcmp(key, "gate_window")) {
        ctx->opt_gate_window = value < 1 ? 1 : value;
    } else if (!strcmp(key, "uniform_fix")) {
        ctx->opt_uniform_fix = value & 63;
    } else if (!strcmp(key, "probe_store")) {
        if (value < 0 || value > 3)
            return rt_fail(ctx, RT_ERR_ARG, "probe_store must be 0..3");
        ctx->opt_probe_store = value;
    } else if (!strcmp(key, "tile_rays")) {
        /* measurement only: tile-major layout (rt_kernels.h, rt_lay); takes
         * effect with the next rt_reserve / rt_set_rays */
        if (value && (value < 64 || value > 65536 || (value & (value - 1))))
            return rt_fail(ctx, RT_ERR_ARG,
                           "tile_rays must be 0 or a power of two in "
                           "[64, 65536]");
        if (value != ctx->opt_tile) {
            RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->opt_tile = value;
            ctx->ld = 0; /* the next rt_reserve lays the arrays out anew */
            ctx->n = 0;
            memset(ctx->valid, 0, sizeof ctx->valid);
        }
    } else if (!strcmp(key, "lds_pad")) {
        /* measurement only: dynamic LDS the kernel never touches, to cap
         * the workgroups resident per CU (160 KB / lds_pad) */
        if (value < 0 || value > 65536)
            return rt_fail(ctx, RT_ERR_ARG, "lds_pad must be in [0, 65536]");
        ctx->opt_lds = value;
    } else if (!strcmp(key, "block")) {
        if (value < 64 || value > 1024 || value % 64)
            return rt_fail(ctx, RT_ERR_ARG, "block must be k*64 in [64,1024]");
        ctx->opt_block = value;
    } else {
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_option: unknown key '%s'", key);
    }
    return RT_OK;
}


int rt_probe(rt_ctx *ctx, int mode, double *ms, double *bytes)
{
    if (!ctx || !ms || !bytes)
        return rt_fail(ctx, RT_ERR_ARG, "rt_probe: NULL argument");
    if (!ctx->d_buf || ctx->nsurf < 2)
        return rt_fail(ctx, RT_ERR_STATE, "rt_probe: set rays first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int L = ctx->buf_nsurf;
    const int64_t ld = ctx->ld;
    if (ctx->opt_tile && (mode >= 1 && mode <= 4))
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_probe: the linear fills address the SoA layout");
    /* rows 1..L-1 of the four arrays; row 0 (the input rays) is preserved */
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    if (mode == 0 || (mode >= 5 && mode <= 9) || mode == 13 || mode == 14) {
        /* the trace kernel's store pattern without its arithmetic, in the
         * layout in force (SoA or tile_rays):
         *   0  80 B/op, 16-byte stores, 48 B/ray input read from HBM
         *   5  same, input from an L2-resident window     6  no input read
         *   7  56 B/op (i served from u), 8-byte stores like the default
         *      kernel, input from HBM                     8  same, no read */
        const int block = ctx->opt_block;
        const int rp = mode >= 7 ? 1 : 2; /* 7..14: one ray per lane */
        const unsigned grid =
            (unsigned)((ld / rp + block - 1) / block);
        const rt_lay lay = rt_layout(ctx);
        const double *win = ctx->d_buf;
        if (mode == 13 || mode == 14) {
            /* 7 with the input rows in their own allocation: 13 = uncached
             * (MTYPE UC: reads bypass the L2), 14 = ordinary device memory */
            const size_t need = (size_t)6 * ld * sizeof(double);
            if (ctx->probe_in_bytes != need || ctx->probe_in_uc != (mode == 13)) {
                if (ctx->d_probe_in)
                    (void)hipFree(ctx->d_probe_in);
                ctx->d_probe_in = NULL;
                ctx->probe_in_bytes = 0;
                if (mode == 13)
                    RT_HIP(ctx, hipExtMallocWithFlags(&ctx->d_probe_in, need,
                                                      hipDeviceMallocUncached));
                else
                    RT_HIP(ctx, hipMalloc(&ctx->d_probe_in, need));
                RT_HIP(ctx, hipMemsetAsync(ctx->d_probe_in, 0, need,
                                           ctx->stream));
                ctx->probe_in_bytes = need;
                ctx->probe_in_uc = mode == 13;
                RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
                RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
            }
            win = (const double *)ctx->d_probe_in;
        }
#define RT_PROBE(IN, RP, SI)                                                  \
    do {                                                                      \
        switch (ctx->opt_probe_store) {                                       \
        case 1: RT_PROBE_FL(IN, RP, SI, 1); break;                            \
        case 2: RT_PROBE_FL(IN, RP, SI, 2); break;                            \
        case 3: RT_PROBE_FL(IN, RP, SI, 3); break;                            \
        default: RT_PROBE_FL(IN, RP, SI, 0); break;                           \
        }                                                                     \
    } while (0)
#define RT_PROBE_FL(IN, RP, SI, FL)                                           \
    hipLaunchKernelGGL((rt_probe_pattern_kernel<IN, RP, FL>), dim3(grid),     \
                       dim3(block), 0, ctx->stream, 1, L, win, lay, ld, SI)
        switch (mode) {
        case 0: RT_PROBE(0, 2, 1); break;
        case 5: RT_PROBE(1, 2, 1); break;
        case 6: RT_PROBE(2, 2, 1); break;
        case 7: RT_PROBE(0, 1, 0); break;
        case 9: RT_PROBE(3, 1, 0); break; /* 7 with non-temporal loads */
        case 13:
        case 14: RT_PROBE(4, 1, 0); break;
        default: RT_PROBE(2, 1, 0); break;
        }
#undef RT_PROBE_FL
#undef RT_PROBE
        *bytes = (double)ld * ((mode >= 7 ? 56. : 80.) * (L - 1) +
                               ((mode == 0 || mode == 7 || mode == 9 ||
                                 mode >= 13) ? 48. : 0.));
    } else if (mode == 10 || mode == 11 || mode == 12) {
        /* 56 B pattern, K = 2 / 4 / 8 rays per lane one after the other,
         * inputs loaded up front */
        const int block = ctx->opt_block;
        const int K = mode == 10 ? 2 : (mode == 11 ? 4 : 8);
        const unsigned grid =
            (unsigned)((ld + (int64_t)block * K - 1) / ((int64_t)block * K));
        const rt_lay lay = rt_layout(ctx);
        if (K == 2)
            hipLaunchKernelGGL(rt_probe_seq_kernel<2>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        else if (K == 4)
            hipLaunchKernelGGL(rt_probe_seq_kernel<4>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        else
            hipLaunchKernelGGL(rt_probe_seq_kernel<8>, dim3(grid), dim3(block),
                               0, ctx->stream, 1, L, lay, ld);
        *bytes = (double)ld * (56. * (L - 1) + 48.);
    } else if (mode == 1) {
        double total = 0.;
        for (int w = RT_Y; w <= RT_T; ++w) {
            const int nc = rt_ncomp(w);
            const int64_t n2 = (int64_t)(L - 1) * nc * ld / 2;
            hipLaunchKernelGGL(rt_probe_fill_kernel, dim3(2048), dim3(256), 0,
                               ctx->stream, rt_arr(ctx, w) + (size_t)nc * ld,
                               n2);
            total += (double)n2 * 16.;
        }
        *bytes = total;
    } else if (mode == 3 || mode == 4) {
        double total = 0.;
        for (int w = RT_Y; w <= RT_T; ++w) {
            const int nc = rt_ncomp(w);
            const int64_t n2 = (int64_t)(L - 1) * nc * ld / 2;
            const unsigned grid = (unsigned)((n2 + 255) / 256);
            if (mode == 3)
                hipLaunchKernelGGL(rt_probe_fill_once_kernel<false>,
                                   dim3(grid), dim3(256), 0, ctx->stream,
                                   rt_arr(ctx, w) + (size_t)nc * ld, n2);
            else
                hipLaunchKernelGGL(rt_probe_fill_once_kernel<true>,
                                   dim3(grid), dim3(256), 0, ctx->stream,
                                   rt_arr(ctx, w) + (size_t)nc * ld, n2);
            total += (double)n2 * 16.;
        }
        *bytes = total;
    } else if (mode == 2) {
        /* copy rows 1..h of Y -> rows 1..h of U, h = L-1: read + write */
        const int64_t n2 = (int64_t)(L - 1) * 3 * ld / 2;
        hipLaunchKernelGGL(rt_probe_copy_kernel, dim3(2048), dim3(256), 0,
                           ctx->stream, rt_arr(ctx, RT_Y) + (size_t)3 * ld,
                           rt_arr(ctx, RT_I) + (size_t)3 * ld, n2);
        *bytes = (double)n2 * 32.;
    } else {
        return rt_fail(ctx, RT_ERR_ARG, "rt_probe: mode %d", mode);
    }
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    RT_HIP(ctx, hipEventSynchronize(ctx->k1));
    float f = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->k0, ctx->k1));
    *ms = f;
    return RT_OK;
}

int rt_download(rt_ctx *ctx, int which, int surf_lo, int surf_hi, double *dst)
{
    if (!ctx || !dst || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download: bad argument");
    if (!ctx->d_buf || surf_lo < 0 || surf_hi > ctx->buf_nsurf ||
        surf_lo >= surf_hi)
        return rt_fail(ctx, RT_ERR_STATE, "rt_download: rows [%d,%d) of %d",
                       surf_lo, surf_hi, ctx->buf_nsurf);
    if (rt_soa_only(ctx, "rt_download") != RT_OK)
        return RT_ERR_STATE;
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


int rt_download_ray(rt_ctx *ctx, int which, int64_t ray, double *dst)
{
    if (!ctx || !dst || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_download_ray: bad argument");
    if (!ctx->d_buf || ray < 0 || ray >= ctx->n)
        return rt_fail(ctx, RT_ERR_STATE, "rt_download_ray: ray %lld of %lld",
                       (long long)ray, (long long)ctx->n);
    if (rt_soa_only(ctx, "rt_download_ray") != RT_OK)
        return RT_ERR_STATE;
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
                                     rt_row(ctx, which, j) + ray,
                                     ctx->ld * sizeof(double), sizeof(double),
                                     nc, hipMemcpyDeviceToHost, ctx->stream));
    }
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_set_weights(rt_ctx *ctx, const double *w)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_set_weights: NULL context");
    if (ctx->n < 1)
        return rt_fail(ctx, RT_ERR_STATE, "rt_set_weights: set rays first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!w) {
        if (ctx->d_w)
            RT_HIP(ctx, hipFree(ctx->d_w));
        ctx->d_w = NULL;
        ctx->w_cap = 0;
        ctx->w_n = 0;
        return RT_OK;
    }
    if ((size_t)ctx->n > ctx->w_cap) {
        if (ctx->d_w)
            RT_HIP(ctx, hipFree(ctx->d_w));
        ctx->d_w = NULL;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_w, ctx->n * sizeof(double)));
        ctx->w_cap = (size_t)ctx->n;
    }
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_w, w, ctx->n * sizeof(double),
                               hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->w_n = ctx->n;
    return RT_OK;
}

/* fetch and add the per-workgroup partials in index order */
static int rt_consumer_ready(rt_ctx *ctx, int surf, const char *who)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "%s: NULL context", who);
    if (!ctx->d_buf || ctx->n < 1 || surf < 0 || surf >= ctx->buf_nsurf ||
        !ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE, "%s: row %d holds no data", who,
                       surf);
    if (rt_soa_only(ctx, who) != RT_OK)
        return RT_ERR_STATE;
    if (ctx->d_w && ctx->w_n != ctx->n)
        return rt_fail(ctx, RT_ERR_STATE,
                       "%s: the weights on the device were set for a batch "
                       "of %lld rays, this one has %lld: call rt_set_weights "
                       "after seeding (NULL for uniform weights)", who,
                       (long long)ctx->w_n, (long long)ctx->n);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_partials)
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_partials,
                              sizeof(double) * (RT_RED_BLOCKS * 8 + 16)));
    return rt_gen_flush(ctx);
}

/* workgroups of a reduction over n rays: no more than there is work for */
static inline unsigned rt_red_blocks(int64_t n)
{
    const int64_t b = (n + RT_RED_THREADS - 1) / RT_RED_THREADS;
    return (unsigned)(b < 1 ? 1 : (b > RT_RED_BLOCKS ? RT_RED_BLOCKS : b));
}

/* device-side second level of a reduction: k sums -> ctx->d_partials tail */
static inline double *rt_reduced(rt_ctx *ctx, int slot)
{
    return ctx->d_partials + (size_t)RT_RED_BLOCKS * 8 + slot;
}

int rt_rms(rt_ctx *ctx, int surf, int64_t ref, double *rms)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_rms");
    if (rc != RT_OK)
        return rc;
    if (!rms || ref >= ctx->n)
        return rt_fail(ctx, RT_ERR_ARG, "rt_rms: bad argument");
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const unsigned blocks = rt_red_blocks(ctx->n);
    /* both passes and their second levels are queued back to back; the host
     * waits once, for one double */
    if (ref < 0) {
        hipLaunchKernelGGL(rt_sum_xy_kernel, dim3(blocks),
                           dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, ctx->n,
                           ctx->ld, ctx->d_partials);
        hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0,
                           ctx->stream, ctx->d_partials, (int)blocks, 2,
                           rt_reduced(ctx, 0));
    }
    hipLaunchKernelGGL(rt_rms_kernel, dim3(blocks), dim3(RT_RED_THREADS), 0,
                       ctx->stream, Yrow, ctx->d_w, 1. / (double)ctx->n,
                       rt_reduced(ctx, 0), ref, ctx->n, ctx->ld,
                       ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 1, rt_reduced(ctx, 2));
    RT_HIP(ctx, hipGetLastError());
    double sum;
    RT_HIP(ctx, hipMemcpyAsync(&sum, rt_reduced(ctx, 2), sizeof sum,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *rms = sqrt(sum);
    return RT_OK;
}

int rt_row_rmax(rt_ctx *ctx, int surf, double *rmax)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_row_rmax");
    if (rc != RT_OK)
        return rc;
    if (!rmax)
        return rt_fail(ctx, RT_ERR_ARG, "rt_row_rmax: NULL");
    hipLaunchKernelGGL(rt_r2max_kernel, dim3(RT_RED_BLOCKS),
                       dim3(RT_RED_THREADS), 0, ctx->stream,
                       rt_row(ctx, RT_Y, surf), ctx->n, ctx->ld,
                       ctx->d_partials);
    double host[RT_RED_BLOCKS * 2];
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(host, ctx->d_partials, sizeof host,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    double mx = 0., bad = 0.;
    for (int b = 0; b < RT_RED_BLOCKS; ++b) {
        mx = host[2 * b] > mx ? host[2 * b] : mx;
        bad = host[2 * b + 1] > bad ? host[2 * b + 1] : bad;
    }
    *rmax = bad ? __builtin_nan("") : sqrt(mx);
    return RT_OK;
}

int rt_spot_stats(rt_ctx *ctx, int surf, int64_t group_rays, int ngroups,
                  double *out)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_spot_stats");
    if (rc != RT_OK)
        return rc;
    if (!out || group_rays < 1 || ngroups < 1 || ngroups > 65535 ||
        group_rays * (int64_t)ngroups != ctx->n)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_spot_stats: %d groups of %lld rays do not tile the "
                       "%lld rays of the batch", ngroups, (long long)group_rays,
                       (long long)ctx->n);
    /* enough workgroups per group to fill the chip, no more than it has rays
     * for */
    int64_t pb = 2048 / ngroups;
    const int64_t fit = (group_rays + RT_RED_THREADS - 1) / RT_RED_THREADS;
    pb = pb > fit ? fit : pb;
    pb = pb < 1 ? 1 : (pb > 256 ? 256 : pb);
    const size_t need = (size_t)ngroups * (RT_GRP_STATS + (size_t)pb * 4);
    if (need > ctx->group_cap) {
        if (ctx->d_group)
            (void)hipFree(ctx->d_group);
        ctx->d_group = nullptr;
        ctx->group_cap = 0;
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_group, need * sizeof(double)));
        ctx->group_cap = need;
    }
    double *stats = ctx->d_group;
    double *partials = stats + (size_t)ngroups * RT_GRP_STATS;
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const dim3 grid((unsigned)pb, (unsigned)ngroups), block(RT_RED_THREADS);
    const dim3 fgrid((unsigned)((ngroups + 63) / 64)), fblock(64);
    hipLaunchKernelGGL(rt_group_sums_kernel, grid, block, 0, ctx->stream, Yrow,
                       ctx->d_w, group_rays, ctx->ld, partials);
    hipLaunchKernelGGL(rt_group_centroid_kernel, fgrid, fblock, 0, ctx->stream,
                       partials, (int)pb, ngroups, stats);
    hipLaunchKernelGGL(rt_group_spread_kernel, grid, block, 0, ctx->stream,
                       Yrow, ctx->d_w, group_rays, ctx->ld, stats, partials);
    hipLaunchKernelGGL(rt_group_finish_kernel, fgrid, fblock, 0, ctx->stream,
                       partials, (int)pb, ngroups, stats);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(out, stats,
                               sizeof(double) * RT_GRP_STATS * ngroups,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return RT_OK;
}

int rt_refocus_shift(rt_ctx *ctx, int surf, double *shift)
{
    int rc = rt_consumer_ready(ctx, surf, "rt_refocus_shift");
    if (rc != RT_OK)
        return rc;
    if (!shift)
        return rt_fail(ctx, RT_ERR_ARG, "rt_refocus_shift: NULL");
    const double *Yrow = rt_row(ctx, RT_Y, surf);
    const double *Irow = rt_row(ctx, RT_I, surf);
    double d[2];
    const unsigned blocks = rt_red_blocks(ctx->n);
    hipLaunchKernelGGL(rt_refocus_sums_kernel, dim3(blocks),
                       dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, Irow, ctx->n,
                       ctx->ld, ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 5, rt_reduced(ctx, 0));
    hipLaunchKernelGGL(rt_refocus_dots_kernel, dim3(blocks),
                       dim3(RT_RED_THREADS), 0, ctx->stream, Yrow, Irow,
                       ctx->d_w, 1. / (double)ctx->n, rt_reduced(ctx, 0),
                       ctx->n, ctx->ld, ctx->d_partials);
    hipLaunchKernelGGL(rt_finalize_kernel, dim3(1), dim3(64), 0, ctx->stream,
                       ctx->d_partials, (int)blocks, 2, rt_reduced(ctx, 5));
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipMemcpyAsync(d, rt_reduced(ctx, 5), sizeof d,
                               hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *shift = -d[0] / d[1];
    return RT_OK;
}

int rt_opd_rays(rt_ctx *ctx, const rt_opd_args *args, double *out_soa)
{
    if (!ctx || !args || !out_soa)
        return rt_fail(ctx, RT_ERR_ARG, "rt_opd_rays: NULL argument");
    int rc = rt_consumer_ready(ctx, 0, "rt_opd_rays");
    if (rc != RT_OK)
        return rc;
    const int L = ctx->buf_nsurf;
    if (args->nrows < 0 || args->nrows > L || args->after < 0 ||
        args->after >= L || args->image < 0 || args->image >= L ||
        args->ref < 0 || args->ref >= ctx->n)
        return rt_fail(ctx, RT_ERR_ARG, "rt_opd_rays: index out of range");
    for (int j = 0; j < L; ++j)
        if (!ctx->valid[j] &&
            (j < args->nrows || j == args->after || j == args->image))
            return rt_fail(ctx, RT_ERR_STATE,
                           "rt_opd_rays: row %d holds no data", j);
    /* reference-ray columns: small strided D2H, then one struct upload */
    rt_opd_ref href;
    memset(&href, 0, sizeof href);
    double col[RT_MAX_SURFACES * 3];
    rc = rt_download_ray(ctx, RT_T, args->ref, col);
    if (rc != RT_OK)
        return rc;
    memcpy(href.t, col, sizeof(double) * L);
    rc = rt_download_ray(ctx, RT_Y, args->ref, col);
    if (rc != RT_OK)
        return rc;
    memcpy(href.y0, col, sizeof(double) * 3);
    memcpy(href.ya, col + 3 * args->after, sizeof(double) * 3);
    memcpy(href.yi, col + 3 * args->image, sizeof(double) * 3);
    rc = rt_download_ray(ctx, RT_U, args->ref, col);
    if (rc != RT_OK)
        return rc;
    memcpy(href.u0, col, sizeof(double) * 3);
    memcpy(href.ua, col + 3 * args->after, sizeof(double) * 3);
    if (!ctx->d_opd_ref)
        RT_HIP(ctx, hipMalloc((void **)&ctx->d_opd_ref, sizeof(rt_opd_ref)));
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_opd_ref, &href, sizeof href,
                               hipMemcpyHostToDevice, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const size_t bytes = (size_t)ctx->n * 3 * sizeof(double);
    rc = rt_need_scratch(ctx, bytes);
    if (rc != RT_OK)
        return rc;
    rc = rt_detach(ctx, RT_U, args->after); /* read at its natural address */
    if (rc != RT_OK)
        return rc;
    const unsigned grid = (unsigned)((ctx->n + 255) / 256);
    RT_HIP(ctx, hipEventRecord(ctx->k0, ctx->stream));
    hipLaunchKernelGGL(rt_opd_kernel, dim3(grid), dim3(256), 0, ctx->stream,
                       *args, ctx->d_opd_ref, rt_arr(ctx, RT_Y),
                       rt_arr(ctx, RT_U), rt_arr(ctx, RT_T), ctx->n, ctx->ld,
                       (double *)ctx->d_scratch);
    RT_HIP(ctx, hipGetLastError());
    RT_HIP(ctx, hipEventRecord(ctx->k1, ctx->stream));
    ctx->traced = 1;
    return rt_d2h(ctx, out_soa, ctx->d_scratch, bytes);
}

int rt_device_ptr(rt_ctx *ctx, int which, int surf, void **out)
{
    if (!ctx || !out || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_device_ptr: bad argument");
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_device_ptr: no such row %d", surf);
    if (!ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE, "rt_device_ptr: row %d holds no data",
                       surf);
    if (rt_soa_only(ctx, "rt_device_ptr") != RT_OK)
        return RT_ERR_STATE;
    int rc = rt_gen_flush(ctx);
    if (rc != RT_OK)
        return rc;
    if (surf == 0)
        ctx->gen_live = 0; /* the caller may write through the pointer */
    *out = rt_row(ctx, which, surf);
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

/* ------------------------------------------------------------------ */
/* multi GPU: RCCL gather of one result row to a root rank            */
/* ------------------------------------------------------------------ */

static int rt_rccl_load(rt_ctx *ctx)
{
    if (g_rccl.lib)
        return RT_OK;
    /* RT_TRANSPORT_LIBRARY names another library with RCCL's entry points:
     * no fallback (if it is set it must load); the GPU tests point it at a
     * shared-memory stand-in so that several ranks sharing one device --
     * which RCCL refuses -- run the nranks > 1 branch below */
    const char *other = getenv("RT_TRANSPORT_LIBRARY");
    void *lib;
    if (other && *other) {
        lib = dlopen(other, RTLD_NOW | RTLD_LOCAL);
        if (!lib)
            return rt_fail(ctx, RT_ERR_RCCL, "dlopen(%s): %s", other,
                           dlerror());
    } else {
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!lib)
            lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!lib)
            return rt_fail(ctx, RT_ERR_RCCL, "dlopen(librccl.so): %s",
                           dlerror());
    }
#define RT_SYM(field, name)                                                   \
    do {                                                                      \
        *(void **)(&g_rccl.field) = dlsym(lib, name);                         \
        if (!g_rccl.field)                                                    \
            return rt_fail(ctx, RT_ERR_RCCL, "dlsym(%s) failed", name);       \
    } while (0)
    RT_SYM(GetUniqueId, "ncclGetUniqueId");
    RT_SYM(CommInitRank, "ncclCommInitRank");
    RT_SYM(CommDestroy, "ncclCommDestroy");
    RT_SYM(GroupStart, "ncclGroupStart");
    RT_SYM(GroupEnd, "ncclGroupEnd");
    RT_SYM(Send, "ncclSend");
    RT_SYM(Recv, "ncclRecv");
    RT_SYM(GetErrorString, "ncclGetErrorString");
#undef RT_SYM
    g_rccl.lib = lib;
    return RT_OK;
}

int rt_comm_unique_id(void *id128)
{
    if (!id128)
        return rt_fail(NULL, RT_ERR_ARG, "rt_comm_unique_id: NULL");
    int rc = rt_rccl_load(NULL);
    if (rc != RT_OK)
        return rc;
    ncclUniqueId id;
    RT_NCCL(NULL, g_rccl.GetUniqueId(&id));
    memcpy(id128, &id, NCCL_UNIQUE_ID_BYTES);
    return RT_OK;
}

int rt_comm_init(rt_ctx *ctx, const void *id128, int nranks, int rank)
{
    if (!ctx || !id128 || nranks < 1 || rank < 0 || rank >= nranks)
        return rt_fail(ctx, RT_ERR_ARG, "rt_comm_init: bad argument");
    if (ctx->comm)
        return rt_fail(ctx, RT_ERR_STATE, "rt_comm_init: already initialised");
    int rc = rt_rccl_load(ctx);
    if (rc != RT_OK)
        return rc;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, NCCL_UNIQUE_ID_BYTES);
    RT_NCCL(ctx, g_rccl.CommInitRank(&ctx->comm, nranks, id, rank));
    ctx->nranks = nranks;
    ctx->rank = rank;
    return RT_OK;
}

int rt_comm_destroy(rt_ctx *ctx)
{
    if (!ctx || !ctx->comm)
        return RT_OK;
    (void)hipStreamSynchronize(ctx->comm_stream);
    g_rccl.CommDestroy(ctx->comm);
    ctx->comm = NULL;
    ctx->nranks = 0;
    return RT_OK;
}

int rt_gather_final(rt_ctx *ctx, int which, int surf, const int64_t *counts,
                    int root, double *d_dst)
{
    if (!ctx || !counts || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "rt_gather_final: bad argument");
    if (!ctx->comm)
        return rt_fail(ctx, RT_ERR_STATE, "rt_gather_final: rt_comm_init first");
    if (root < 0 || root >= ctx->nranks)
        return rt_fail(ctx, RT_ERR_ARG, "rt_gather_final: root %d", root);
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "rt_gather_final: no row %d", surf);
    if (!ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_gather_final: row %d holds no data", surf);
    if (counts[ctx->rank] != ctx->n)
        return rt_fail(ctx, RT_ERR_ARG,
                       "rt_gather_final: counts[%d]=%lld but this rank holds "
                       "%lld rays",
                       ctx->rank, (long long)counts[ctx->rank],
                       (long long)ctx->n);
    if (ctx->rank == root && !d_dst)
        return rt_fail(ctx, RT_ERR_ARG, "rt_gather_final: root needs d_dst");
    if (rt_soa_only(ctx, "rt_gather_final") != RT_OK)
        return RT_ERR_STATE;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }

    const int nc = rt_ncomp(which);
    const int p = ctx->parity;
    ctx->parity ^= 1;
    const size_t row_bytes = (size_t)nc * ctx->n * sizeof(double);
    if (row_bytes > ctx->stage_bytes) {
        RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
        RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
        for (int i = 0; i < 2; ++i) {
            if (ctx->d_stage[i])
                RT_HIP(ctx, hipFree(ctx->d_stage[i]));
            ctx->d_stage[i] = NULL;
            RT_HIP(ctx, hipMalloc((void **)&ctx->d_stage[i], row_bytes));
            ctx->gather_pending[i] = 0;
        }
        ctx->stage_bytes = row_bytes;
    }

    /* trace stream: snapshot the row (compact, ld -> n) into stage[p], so the
     * next trace may overwrite the row while RCCL is still sending it.  The
     * snapshot of step k+2 must wait for the gather of step k. */
    if (ctx->gather_pending[p])
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gathered[p], 0));
    const double *src = rt_row(ctx, which, surf);
    RT_HIP(ctx, hipMemcpy2DAsync(ctx->d_stage[p], ctx->n * sizeof(double), src,
                                 ctx->ld * sizeof(double),
                                 ctx->n * sizeof(double), nc,
                                 hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->staged[p], ctx->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->staged[p], 0));

    int64_t total = 0, my_off = 0;
    for (int r = 0; r < ctx->nranks; ++r) {
        if (r == ctx->rank)
            my_off = total;
        total += counts[r];
    }
    const double *stage = ctx->d_stage[p];
    if (ctx->rank == root) {
        /* own shard: device-to-device, no RCCL */
        for (int c = 0; c < nc; ++c)
            RT_HIP(ctx, hipMemcpyAsync(d_dst + (size_t)c * total + my_off,
                                       stage + (size_t)c * ctx->n,
                                       ctx->n * sizeof(double),
                                       hipMemcpyDeviceToDevice,
                                       ctx->comm_stream));
    }
    if (ctx->nranks > 1) {
        RT_NCCL(ctx, g_rccl.GroupStart());
        if (ctx->rank == root) {
            int64_t off = 0;
            for (int r = 0; r < ctx->nranks; ++r) {
                if (r != root) {
                    for (int c = 0; c < nc; ++c)
                        RT_NCCL(ctx, g_rccl.Recv(d_dst + (size_t)c * total + off,
                                                 (size_t)counts[r], ncclDouble,
                                                 r, ctx->comm,
                                                 ctx->comm_stream));
                }
                off += counts[r];
            }
        } else {
            for (int c = 0; c < nc; ++c)
                RT_NCCL(ctx, g_rccl.Send(stage + (size_t)c * ctx->n,
                                         (size_t)ctx->n, ncclDouble, root,
                                         ctx->comm, ctx->comm_stream));
        }
        RT_NCCL(ctx, g_rccl.GroupEnd());
    }
    RT_HIP(ctx, hipEventRecord(ctx->gathered[p], ctx->comm_stream));
    ctx->gather_pending[p] = 1;
    return RT_OK;
}

int rt_comm_sync(rt_ctx *ctx)
{
    if (!ctx)
        return rt_fail(ctx, RT_ERR_ARG, "rt_comm_sync: NULL context");
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->comm_stream));
    return RT_OK;
}

} /* extern "C" */
