/*
 * rt_ctx.h -- what the translation units of librt_mi355.so share: the
 * context, error plumbing, row addressing and the few internal helpers that
 * cross TU borders.  Not installed; the public surface is include/rt_mi355.h.
 *
 *   rt_engine.hip     contexts, surface tables, seeding / generation, the
 *                     trace launch, row bookkeeping (served rows), downloads
 *   rt_consumers.hip  aiming kernel, rms / refocus / spot statistics / opd
 *   rt_comm.hip       RCCL communicator and the gather of a result row
 */
#ifndef RT_CTX_H
#define RT_CTX_H

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>

#include "rt_math.h"
#include "rt_lay.h"

static inline double rt_now_ms(void)
{
    return std::chrono::duration<double, std::milli>(
               std::chrono::steady_clock::now().time_since_epoch()).count();
}

#define RT_INTERNAL __attribute__((visibility("hidden")))

#define RT_NEVENTS 8
#define RT_MAX_GROUPS 65535 /* surface tables per launch (wavelengths, or
                               variants of a system: tolerancing runs) */
#define RT_GATHER_SLOTS 4   /* staging buffers of rt_gather_final: gathers of
                               that many steps (or chunks) may be in flight */

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
    ncclResult_t (*CommCount)(const ncclComm_t, int *);
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *);
    ncclResult_t (*GetVersion)(int *);
};


/*
 * Where the result arrays live (rt_place.h): pieces of device memory mapped
 * behind one address range; the range is kept only if the batch's own store
 * pattern runs at the fast level behind it.
 */
/* batches whose planes pass RT_BLOCK_ONE bytes are cut into blocks of at most
 * RT_BLOCK_BYTES (rt_lay.h).  Measured (C3, per 10^7 rays): one block of
 * 8.7 GB 0.978 ms = two of 4.4; one block of 10.4 GB (10^7 rays) 1.007, two
 * of 5.2 GB 0.994; one of 20.8 GB 1.18-1.19, three of 6.9 GB 0.96-1.07;
 * blocks of 5.2 ... 8.7 GB all alike */
#define RT_BLOCK_ONE 8.5e9
#define RT_BLOCK_BYTES 7.0e9

/* generated batches whose pupil points (16 B each, read once per bundle) pass
 * RT_TURN_ABOVE bytes are traced in turns of RT_TURN_POINTS points
 * (rt_gen_wg, rt_lay.h): what is read again should still be in the 256 MB
 * Infinity Cache */
#define RT_TURN_ABOVE (128. * 1048576.)
#define RT_TURN_POINTS (4 * 1048576)

#define RT_PLACE_CLASSES 4
struct rt_place {
    void *base;      /* the mapped range (= d_buf), NULL: plain hipMalloc */
    size_t bytes, piece;
    int n;           /* pieces mapped */
    void *handles;   /* hipMemGenericAllocationHandle_t[n] */
    int created;     /* pieces created on the way (the surplus was released) */
    int ballast;     /* blocks of ballast held while searching */
    int nclass;      /* classes seen */
    int count[RT_PLACE_CLASSES]; /* pieces of each class among the n */
    int mixed;       /* >= a third of the pieces outside the largest class */
    int class_mix;   /* (the same: what the classes alone said) */
    float self_ms, cross_ms; /* pair test: same piece / another class */
    float store_gbps; /* the batch's own store pattern over the arrays
                         (0: not measured) */
    int fast;         /* four workgroups per CU: store_gbps at the fast level
                         (where nothing was measured: the classes are mixed) */
    /* wall time of the search (rt_place_alloc): all of it, the pieces
     * (create, map, pair tests), the ballast, unmapping / releasing / the
     * final mapping; and of measuring / re-mapping (rt_place_tune) */
    float search_ms, pieces_ms, ballast_ms, remap_ms, tune_ms;
    int picks;           /* sets of pieces tried (rt_place_settle) */
    float pick_gbps[5];  /* the batch's store pattern over each */
    float slowest_create_ms; /* the longest single hipMemCreate of the search */
    int cut_short;       /* the search ended early: 1 = its time budget was
                            up, 2 = a hipMemCreate stalled */
    unsigned char slot_cls[64]; /* class of the piece in each slot of the range
                                   (the first 64; the log's) */
    int orders;          /* other orders of a set's pieces along the range that
                            were mapped and measured (rt_place_orders) */
    int settled;         /* rt_place_settle has searched for these arrays: a
                            later layout in the same buffer only measures */
    float total_ms;      /* wall time of everything rt_reserve spent on the
                            placement of this buffer (all sets, measurements,
                            the coherence proof) */
};

struct rt_ctx {
    int device;
    hipStream_t stream;      /* trace + copies */
    hipStream_t comm_stream; /* RCCL gather */
    hipStream_t copy_stream; /* device -> host DMAs of row downloads (rt_d2h_jobs);
                                the seed kernels of a windowed upload */
    hipEvent_t seed_ev;      /* orders those against the uploads */
    int pin_next;            /* staging buffer the next upload chunk takes */
    float uni_share;         /* share of the launch components that is uniform
                                across a tile, sampled (-1: not known) */
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

    unsigned *d_uni; /* per 64-ray tile of row 0 (seed kernels): note[cap]
                        -- which launch components are one bit pattern across
                        the tile -- then first[6][cap], the tile's first ray */
    size_t uni_cap;  /* tiles d_uni holds */
    int uni_valid;   /* d_uni describes what row 0 holds now */
    int opt_uniform; /* the trace uses it (default on) */
    double *d_buf; /* Y | U | I | T */
    size_t cap_doubles;
    int64_t n, ld;
    /* the batch in blocks (rt_lay.h): ld = nblk * bs ray slots, rows bs
     * doubles apart, bts doubles from block to block (0: one block, bs = ld) */
    int64_t bs, bts;
    int nblk;
    int opt_block; /* rays per block asked for (0: chosen by rt_reserve) */
    int opt_pitch; /* row pitch rounded up to a multiple of this many rays */
    int64_t opt_turn; /* pupil points per turn of a generated batch (rt_gen_wg):
                         0 automatic, -1 never */
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
    double *d_partials; /* RT_RED_BLOCKS x 16 doubles + 16 reduced values */
    double *h_res; /* pinned, 8 doubles: where a consumer's last kernel writes
                    * its scalars (no D2H copy after it) */
    double *d_group;    /* rt_spot_stats: stats | partials */
    size_t group_cap;   /* doubles */
    double *h_rows;     /* pinned: rt_row_stats' results, then the ticket the
                         * finishing kernel signs (the host spins on it) */
    unsigned *d_arrived; /* bundles whose statistics are written (one word) */
    unsigned long long row_seq; /* calls of rt_row_stats so far */
    double *h_group;    /* pinned: the stats as rt_group_finish_kernel writes
                         * them (up to RT_GROUP_PINNED groups) */
    struct rt_opd_ref *d_opd_ref; /* reference-ray columns, one per bundle */
    size_t opd_ref_cap;
    double *d_opd;   /* x | y | t of the last rt_opd_rays / rt_opd_stats(keep) */
    size_t opd_cap;  /* doubles */
    int64_t opd_n;   /* rays it holds (0: nothing) */
    double *h_opd;   /* pinned: where rt_opd_finish_kernel writes the stats */

    /* kernel choices */
    int opt_alias;
    int opt_fuse; /* build generated rays inside the first trace */
    int opt_fast; /* aspheric elements on the fast arithmetic (RT_F_FAST) */
    int opt_range; /* quotients / roots without range scaffolding (RT_F_RANGE) */
    int opt_onepass; /* rms / refocus sums in one pass over the rows */
    int opt_cevents; /* the consumers bracket their kernels with k0 / k1 */
    int opt_resident; /* bytes of unused dynamic LDS per workgroup of the
                         trace kernels: caps the workgroups resident per CU
                         (160 KB / bytes); -1 = chosen per trace */
    /* a step traced in pieces (rt_trace_chunk): until every piece has been
     * traced the rows hold new and old columns side by side, and whatever
     * reads whole rows must wait (rt_rows_whole) */
    int pieces_total, pieces_seen, pieces_start, pieces_stop, pieces_clip;
    uint64_t pieces_mask[4];
    int gather_seen, gather_nchunks; /* chunks of the gather in progress */
    int opt_place;    /* large arrays in class-mixed pieces (rt_place.h) */
    int opt_place_orders; /* orders of a set's pieces tried (-1: 6 for arrays
                             below 4 GiB, 3 up to 16 GiB) */
    float opt_place_budget_ms; /* wall time an allocation may spend choosing
                                  memory (rt_place.h: RT_PLACE_BUDGET_MS) */
    float opt_place_good; /* GB/s of the store pattern at which the search
                             for a better range / set of pieces ends */
    struct rt_place place;
    /* the layout rt_reserve is about to make (rt_place_weights) */
    long long plan_L, plan_bs, plan_nblk;
    double place_deadline_ms; /* steady clock: when the current allocation's
                                 search for better memory ends (0: none
                                 running) */
    int place_incoherent; /* a placement of this context failed its check
                             (rt_place_coherent) and was given up */
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
    int table_asph; /* some element of some table is aspheric: the trace
                       kernels with the Newton solves (else the lean ones) */

    /* multi GPU (rt_comm.hip) */
    ncclComm_t comm;
    int nranks, rank;
    double *d_stage[RT_GATHER_SLOTS];
    size_t stage_bytes[RT_GATHER_SLOTS];
    hipEvent_t staged[RT_GATHER_SLOTS], gathered[RT_GATHER_SLOTS];
    hipEvent_t g0, g1; /* timing: first / last operation of a gather on the
                          communication stream */
    int gather_pending[RT_GATHER_SLOTS];
    int gather_slot;
    int gather_timed; /* g0 and g1 have both been recorded */

    char err[512];
};

RT_INTERNAL extern rt_rccl_api g_rccl;
RT_INTERNAL extern int g_place_distrust; /* placement given up process-wide */

extern "C" RT_INTERNAL int rt_fail(rt_ctx *ctx, int code, const char *fmt, ...);

/* refuse to read whole rows in the middle of a step traced in pieces */
#define RT_ROWS_WHOLE(ctx, who)                                               \
    do {                                                                      \
        if ((ctx)->pieces_seen > 0)                                           \
            return rt_fail(ctx, RT_ERR_STATE,                                 \
                           "%s: %d of %d pieces of the current step have "    \
                           "been traced (rt_trace_chunk): the rows are not "  \
                           "whole yet", who, (ctx)->pieces_seen,              \
                           (ctx)->pieces_total);                              \
    } while (0)

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
    /* block 0: Y,U,I are [L][3][bs]; T is [L][bs] */
    const size_t plane = (size_t)c->buf_nsurf * 3 * (size_t)c->bs;
    return c->d_buf + (size_t)which * plane;
}

static inline int rt_ncomp(int which) { return which == RT_T ? 1 : 3; }

/* addressing of the result arrays as the kernels see it (rt_lay.h) */
static inline rt_lay rt_layout_planes(const rt_ctx *c)
{
    rt_lay a;
    a.j0 = 0;
    a.wgs = a.wmagic = a.wshift = a.w0 = 0;
    a.Y = rt_arr(c, RT_Y);
    a.U = rt_arr(c, RT_U);
    a.I = rt_arr(c, RT_I);
    a.T = rt_arr(c, RT_T);
    a.cs = c->bs;
    a.ss = 3 * c->bs;
    a.ssT = c->bs;
    a.bs = c->bs;
    a.ts = c->bts;
    return a;
}

/* ... of the whole batch (a window of it: rt_lay_set_window afterwards) */
static inline rt_lay rt_layout(const rt_ctx *c)
{
    rt_lay a = rt_layout_planes(c);
    rt_lay_set_window(a, 0, c->ld);
    return a;
}

/* device address of one surface row, resolving the I -> U aliasing */
static inline double *rt_row(const rt_ctx *c, int which, int surf)
{
    if (which == RT_I && c->i_alias[surf])
        return rt_row(c, RT_U, c->i_alias[surf] == 1 ? surf - 1 : surf);
    if (which == RT_U && c->u_alias[surf])
        return rt_row(c, RT_I, surf); /* surf >= 1, and I[surf] never points
                                         back at U[surf] there */
    return rt_arr(c, which) + (size_t)surf * rt_ncomp(which) * c->bs;
}

/*
 * How rt_reserve cuts a batch of nrays rays through nsurf elements: one block
 * (the documented layout, rows `quantum`-padded) as long as the planes of the
 * batch stay below RT_BLOCK_ONE bytes; above it the fewest blocks of at most
 * RT_BLOCK_BYTES, all of the same whole number of 256-ray workgroups
 * (rt_lay.h).  opt_block > 0: blocks of about that many rays instead (tests).
 * A laboratory layout (quantum != 64) has the arrays to itself.
 */
static inline void rt_block_plan(int nsurf, int64_t opt_block, int64_t quantum,
                                 int64_t nrays, int64_t *bs, int *nblk)
{
    *bs = (nrays + quantum - 1) / quantum * quantum;
    *nblk = 1;
    const double bytes = 80. * nsurf * (double)nrays;
    int64_t want = 1; /* blocks asked for */
    if (quantum != 64)
        want = 1;
    else if (opt_block > 0)
        want = (nrays + opt_block - 1) / opt_block;
    else if (bytes > RT_BLOCK_ONE)
        want = (int64_t)ceil(bytes / RT_BLOCK_BYTES);
    if (want > 1) {
        const int64_t b = ((nrays + want - 1) / want + 255) / 256 * 256;
        const int64_t k = (nrays + b - 1) / b;
        if (k >= 2) { /* (else rounding up to whole workgroups ate a block) */
            *bs = b;
            *nblk = (int)k;
        }
    }
}

/* the rays [lo, hi) of a row as contiguous segments, block by block:
 * for (rt_seg s = rt_seg_first(c, lo, hi); s.cnt; s = rt_seg_next(c, s, hi))
 * -- s.off doubles from the row's start in block 0, s.ray the first ray */
struct rt_seg {
    int64_t ray, cnt, off;
};

static inline rt_seg rt_seg_at(const rt_ctx *c, int64_t ray, int64_t hi)
{
    rt_seg s = {ray, 0, 0};
    if (ray >= hi)
        return s;
    const int64_t b = c->bts ? ray / c->bs : 0;
    const int64_t end = c->bts ? (b + 1) * c->bs : hi;
    s.cnt = (end < hi ? end : hi) - ray;
    s.off = rt_block_col(c->bs, c->bts, ray);
    return s;
}
#define RT_FOR_SEGMENTS(c, s, lo, hi)                                       \
    for (rt_seg s = rt_seg_at(c, lo, hi); s.cnt;                            \
         s = rt_seg_at(c, s.ray + s.cnt, hi))

extern "C" { /* (linkage only: none of these is exported) */
/* rt_engine.hip */
RT_INTERNAL int rt_detach(rt_ctx *c, int which, int surf);
RT_INTERNAL int rt_gen_flush(rt_ctx *c);
RT_INTERNAL int rt_need_scratch(rt_ctx *ctx, size_t bytes);
RT_INTERNAL int rt_h2d(rt_ctx *ctx, void *dst, const void *src, size_t bytes);
RT_INTERNAL int rt_d2h(rt_ctx *ctx, void *dst, const void *src, size_t bytes);
/* rt_comm.hip */
RT_INTERNAL void rt_comm_release(rt_ctx *ctx);
}


#endif /* RT_CTX_H */
