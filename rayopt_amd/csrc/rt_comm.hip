/*
 * rt_comm.hip -- multi GPU: one process per GPU, ray shards, no exchange in
 * the trace.  The only exchange of a job is the gather of one result row (the
 * last-surface intercepts) to a root rank: grouped ncclSend / ncclRecv over
 * the peers' direct xGMI links to the root (no ring: xGMI is point to point
 * and the root's ingest over its 7 links is the bound).  The row is
 * snapshotted compactly (ld -> n) into a staging buffer on the trace stream
 * and sent on a second stream, so the next trace -- or the next CHUNK of the
 * same trace, rt_gather_chunk -- runs while RCCL moves the previous one.
 *
 * The reference has nothing here (no NCCL / MPI / Gloo);
 * this is BASELINE.json's "RCCL gather over xGMI only for the final intercept
 * arrays".  Part of librt_mi355.so; librccl.so is bound at first use.
 */
#include <dlfcn.h>

#include "rt_ctx.h"

rt_rccl_api g_rccl = {};

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
    rt_rccl_api api = {};
#define RT_SYM(field, name)                                                   \
    do {                                                                      \
        *(void **)(&api.field) = dlsym(lib, name);                            \
        if (!api.field) {                                                     \
            dlclose(lib);                                                     \
            return rt_fail(ctx, RT_ERR_RCCL, "dlsym(%s) failed", name);       \
        }                                                                     \
    } while (0)
    RT_SYM(GetUniqueId, "ncclGetUniqueId");
    RT_SYM(CommInitRank, "ncclCommInitRank");
    RT_SYM(CommDestroy, "ncclCommDestroy");
    RT_SYM(GroupStart, "ncclGroupStart");
    RT_SYM(GroupEnd, "ncclGroupEnd");
    RT_SYM(Send, "ncclSend");
    RT_SYM(Recv, "ncclRecv");
    RT_SYM(GetErrorString, "ncclGetErrorString");
    RT_SYM(CommCount, "ncclCommCount");
    RT_SYM(CommUserRank, "ncclCommUserRank");
    RT_SYM(GetVersion, "ncclGetVersion");
#undef RT_SYM
    api.lib = lib;
    g_rccl = api;
    return RT_OK;
}

/* the staging buffers (rt_destroy, and when they have to grow) */
void rt_comm_release(rt_ctx *ctx)
{
    for (int i = 0; i < RT_GATHER_SLOTS; ++i) {
        if (ctx->d_stage[i])
            (void)hipFree(ctx->d_stage[i]);
        ctx->d_stage[i] = NULL;
        ctx->stage_bytes[i] = 0;
        ctx->gather_pending[i] = 0;
    }
}

/* slot `p` large enough for `bytes`: what a slot can hold is recorded per
 * slot and only after its allocation succeeded, so a failed hipMalloc leaves
 * an empty slot behind, never a size without a buffer */
static int rt_stage_reserve(rt_ctx *ctx, int p, size_t bytes)
{
    if (bytes <= ctx->stage_bytes[p])
        return RT_OK;
    if (ctx->gather_pending[p]) /* RCCL may still read the old buffer */
        RT_HIP(ctx, hipEventSynchronize(ctx->gathered[p]));
    ctx->gather_pending[p] = 0;
    if (ctx->d_stage[p])
        (void)hipFree(ctx->d_stage[p]);
    ctx->d_stage[p] = NULL;
    ctx->stage_bytes[p] = 0;
    hipError_t e = hipMalloc((void **)&ctx->d_stage[p], bytes);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ctx->d_stage[p] = NULL;
        return rt_fail(ctx, RT_ERR_NOMEM,
                       "rt_gather_final: staging hipMalloc(%zu): %s", bytes,
                       hipGetErrorString(e));
    }
    ctx->stage_bytes[p] = bytes;
    return RT_OK;
}

/* rows [lo_r, hi_r) of every rank r: what rank r holds of chunk `chunk` */
static int rt_gather_window(rt_ctx *ctx, int which, int surf,
                            const int64_t *counts, int root, double *d_dst,
                            int chunk, int nchunks, const char *who)
{
    if (!ctx || !counts || which < RT_Y || which > RT_T)
        return rt_fail(ctx, RT_ERR_ARG, "%s: bad argument", who);
    if (!ctx->comm)
        return rt_fail(ctx, RT_ERR_STATE, "%s: rt_comm_init first", who);
    if (root < 0 || root >= ctx->nranks)
        return rt_fail(ctx, RT_ERR_ARG, "%s: root %d of %d ranks", who, root,
                       ctx->nranks);
    if (!ctx->d_buf || surf < 0 || surf >= ctx->buf_nsurf)
        return rt_fail(ctx, RT_ERR_STATE, "%s: no row %d", who, surf);
    if (!ctx->valid[surf])
        return rt_fail(ctx, RT_ERR_STATE, "%s: row %d holds no data", who,
                       surf);
    if (counts[ctx->rank] != ctx->n)
        return rt_fail(ctx, RT_ERR_ARG,
                       "%s: counts[%d]=%lld but this rank holds %lld rays",
                       who, ctx->rank, (long long)counts[ctx->rank],
                       (long long)ctx->n);
    for (int r = 0; r < ctx->nranks; ++r)
        if (counts[r] < 0)
            return rt_fail(ctx, RT_ERR_ARG, "%s: counts[%d] < 0", who, r);
    if (ctx->rank == root && !d_dst)
        return rt_fail(ctx, RT_ERR_ARG, "%s: root needs d_dst", who);
    if (nchunks == 1) /* a whole row: not in the middle of a step in pieces */
        RT_ROWS_WHOLE(ctx, who);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    {
        int rc = rt_gen_flush(ctx);
        if (rc != RT_OK)
            return rc;
    }
    int64_t lo, hi;
    if (rt_chunk_bounds(ctx->n, chunk, nchunks, &lo, &hi) != RT_OK)
        return rt_fail(ctx, RT_ERR_ARG, "%s: chunk %d of %d", who, chunk,
                       nchunks);
    const int64_t mine = hi - lo;

    const int nc = rt_ncomp(which);
    const int p = ctx->gather_slot;
    ctx->gather_slot = (p + 1) % RT_GATHER_SLOTS;
    {
        int rc = rt_stage_reserve(ctx, p, (size_t)nc * (mine > 0 ? mine : 1) *
                                              sizeof(double));
        if (rc != RT_OK)
            return rc;
    }

    /* trace stream: snapshot the rows (compact, ld -> count) into stage[p],
     * so the next trace may overwrite the row while RCCL is still sending
     * it.  The snapshot must wait for the gather that used this slot last. */
    if (ctx->gather_pending[p])
        RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->gathered[p], 0));
    const double *src = rt_row(ctx, which, surf);
    RT_FOR_SEGMENTS(ctx, g, lo, lo + mine) /* block by block of the batch */
        RT_HIP(ctx, hipMemcpy2DAsync(ctx->d_stage[p] + (g.ray - lo),
                                     mine * sizeof(double), src + g.off,
                                     ctx->bs * sizeof(double),
                                     (size_t)g.cnt * sizeof(double), nc,
                                     hipMemcpyDeviceToDevice, ctx->stream));
    RT_HIP(ctx, hipEventRecord(ctx->staged[p], ctx->stream));
    RT_HIP(ctx, hipStreamWaitEvent(ctx->comm_stream, ctx->staged[p], 0));
    /* the gather's clock runs from the first piece ISSUED to the last one
     * issued, whatever their order */
    if (ctx->gather_seen == 0 || ctx->gather_nchunks != nchunks) {
        ctx->gather_seen = 0;
        ctx->gather_nchunks = nchunks;
        ctx->gather_timed = 0;
        RT_HIP(ctx, hipEventRecord(ctx->g0, ctx->comm_stream));
    }

    int64_t total = 0;
    for (int r = 0; r < ctx->nranks; ++r)
        total += counts[r];
    const double *stage = ctx->d_stage[p];
    /* destination on root: [component][global ray]; rank r's rays start at
     * sum(counts[:r]), its chunk at + lo_r */
    if (ctx->rank == root && mine > 0) {
        int64_t my_off = 0;
        for (int r = 0; r < ctx->rank; ++r)
            my_off += counts[r];
        /* own shard: device-to-device, no RCCL */
        RT_HIP(ctx, hipMemcpy2DAsync(d_dst + my_off + lo,
                                     total * sizeof(double), stage,
                                     mine * sizeof(double),
                                     mine * sizeof(double), nc,
                                     hipMemcpyDeviceToDevice,
                                     ctx->comm_stream));
    }
    if (ctx->nranks > 1) {
        RT_NCCL(ctx, g_rccl.GroupStart());
        ncclResult_t bad = ncclSuccess;
        if (ctx->rank == root) {
            int64_t off = 0;
            for (int r = 0; r < ctx->nranks && bad == ncclSuccess; ++r) {
                int64_t rlo, rhi;
                (void)rt_chunk_bounds(counts[r], chunk, nchunks, &rlo, &rhi);
                if (r != root && rhi > rlo)
                    for (int c = 0; c < nc && bad == ncclSuccess; ++c)
                        bad = g_rccl.Recv(d_dst + (size_t)c * total + off + rlo,
                                          (size_t)(rhi - rlo), ncclDouble, r,
                                          ctx->comm, ctx->comm_stream);
                off += counts[r];
            }
        } else if (mine > 0) {
            for (int c = 0; c < nc && bad == ncclSuccess; ++c)
                bad = g_rccl.Send(stage + (size_t)c * mine, (size_t)mine,
                                  ncclDouble, root, ctx->comm,
                                  ctx->comm_stream);
        }
        /* a group that was opened is always closed, also after a failed
         * call inside it */
        const ncclResult_t end = g_rccl.GroupEnd();
        if (bad != ncclSuccess || end != ncclSuccess)
            return rt_fail(ctx, RT_ERR_RCCL, "%s: %s", who,
                           g_rccl.GetErrorString(bad != ncclSuccess ? bad
                                                                    : end));
    }
    RT_HIP(ctx, hipEventRecord(ctx->gathered[p], ctx->comm_stream));
    if (++ctx->gather_seen >= nchunks) {
        RT_HIP(ctx, hipEventRecord(ctx->g1, ctx->comm_stream));
        ctx->gather_timed = 1;
        ctx->gather_seen = 0;
    }
    ctx->gather_pending[p] = 1;
    return RT_OK;
}

extern "C" {

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
    /* what the communicator itself says it is */
    int seen = -1, me = -1;
    ncclResult_t r1 = g_rccl.CommCount(ctx->comm, &seen);
    ncclResult_t r2 = g_rccl.CommUserRank(ctx->comm, &me);
    if (r1 != ncclSuccess || r2 != ncclSuccess || seen != nranks ||
        me != rank) {
        g_rccl.CommDestroy(ctx->comm);
        ctx->comm = NULL;
        return rt_fail(ctx, RT_ERR_RCCL,
                       "rt_comm_init: asked for rank %d of %d, the "
                       "communicator says rank %d of %d", rank, nranks, me,
                       seen);
    }
    ctx->nranks = nranks;
    ctx->rank = rank;
    ctx->gather_seen = 0;
    return RT_OK;
}

int rt_comm_info(rt_ctx *ctx, int info[4], int *link_type, int *hops,
                 int max_devices)
{
    if (!ctx || !info || max_devices < 0 ||
        (max_devices > 0 && (!link_type || !hops)))
        return rt_fail(ctx, RT_ERR_ARG, "rt_comm_info: bad argument");
    if (!ctx->comm)
        return rt_fail(ctx, RT_ERR_STATE, "rt_comm_info: rt_comm_init first");
    info[0] = info[1] = info[2] = -1;
    RT_NCCL(ctx, g_rccl.CommCount(ctx->comm, &info[0]));
    RT_NCCL(ctx, g_rccl.CommUserRank(ctx->comm, &info[1]));
    RT_NCCL(ctx, g_rccl.GetVersion(&info[2]));
    int ndev = 0;
    RT_HIP(ctx, hipGetDeviceCount(&ndev));
    info[3] = ndev;
    for (int d = 0; d < ndev && d < max_devices; ++d) {
        uint32_t type = 0, hop = 0;
        link_type[d] = hops[d] = -1; /* this device itself, or no answer */
        if (d != ctx->device &&
            hipExtGetLinkTypeAndHopCount(ctx->device, d, &type, &hop) ==
                hipSuccess) {
            link_type[d] = (int)type;
            hops[d] = (int)hop;
        }
        (void)hipGetLastError();
    }
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
    return rt_gather_window(ctx, which, surf, counts, root, d_dst, 0, 1,
                            "rt_gather_final");
}

int rt_gather_chunk(rt_ctx *ctx, int which, int surf, const int64_t *counts,
                    int root, double *d_dst, int chunk, int nchunks)
{
    if (nchunks < 1 || chunk < 0 || chunk >= nchunks)
        return rt_fail(ctx, RT_ERR_ARG, "rt_gather_chunk: chunk %d of %d",
                       chunk, nchunks);
    return rt_gather_window(ctx, which, surf, counts, root, d_dst, chunk,
                            nchunks, "rt_gather_chunk");
}

int rt_gather_ms(rt_ctx *ctx, double *total_ms, double *exposed_ms)
{
    if (!ctx || !total_ms || !exposed_ms)
        return rt_fail(ctx, RT_ERR_ARG, "rt_gather_ms: NULL argument");
    if (!ctx->comm || !ctx->traced || !ctx->gather_timed)
        return rt_fail(ctx, RT_ERR_STATE,
                       "rt_gather_ms: no complete gather yet (the last chunk "
                       "closes it)");
    RT_HIP(ctx, hipEventSynchronize(ctx->g1));
    RT_HIP(ctx, hipEventSynchronize(ctx->k1));
    float f = 0.f;
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->g0, ctx->g1));
    *total_ms = f;
    /* what the exchange adds behind the last trace kernel of this rank */
    RT_HIP(ctx, hipEventElapsedTime(&f, ctx->k1, ctx->g1));
    *exposed_ms = f > 0.f ? f : 0.;
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
