/*
 * shm_transport.cpp -- TEST INFRASTRUCTURE, not part of the product.
 *
 * A stand-in for librccl.so that exports the eight nccl* entry points
 * librt_mi355.so binds at run time (rt_engine.hip: rt_rccl_load) and moves the
 * data through POSIX shared memory (/dev/shm) instead of xGMI.  RCCL refuses
 * two ranks on one device ("Duplicate GPU detected"), and the build boxes of
 * this project have one GPU: with RT_TRANSPORT_LIBRARY pointing here, N
 * processes that share ONE device run the engine's real nranks > 1 gather --
 * staging, offsets, counts, grouped send/recv calls, the double-buffered
 * pipeline -- on real device rows.  What it cannot exercise is RCCL itself.
 *
 * Semantics kept from NCCL: point-to-point messages between a pair of ranks
 * match in the order they were issued; calls between ncclGroupStart and
 * ncclGroupEnd are deferred to the end of the group (all sends, then all
 * receives, so a group never deadlocks); work is ordered after what is already
 * queued on the stream (here: by synchronising it).  Everything is synchronous.
 *
 * A message from rank s to rank d with sequence number q is the file
 * $RT_SHM_TRANSPORT_DIR (default /dev/shm)/<tag>.<s>.<d>.<q>, written under a temporary name and renamed
 * (atomic publication); the receiver polls for it, copies it to the device and
 * unlinks it.
 */
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace {

constexpr int MAX_RANKS = 64;

struct comm {
    int nranks, rank;
    char tag[96];
    uint64_t sent[MAX_RANKS], received[MAX_RANKS];
};

struct op {
    bool send;
    void *buf;
    size_t bytes;
    int peer;
    comm *c;
    hipStream_t stream;
};

thread_local int group_depth = 0;
thread_local std::vector<op> pending;

size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}

double now()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void name_of(char *out, size_t cap, const comm *c, int src, int dst,
             uint64_t seq, const char *suffix)
{
    const char *dir = getenv("RT_SHM_TRANSPORT_DIR");
    snprintf(out, cap, "%s/%s.%d.%d.%llu%s", dir && *dir ? dir : "/dev/shm",
             c->tag, src, dst, (unsigned long long)seq, suffix);
}

ncclResult_t do_send(const op &o)
{
    comm *c = o.c;
    char tmp[512], path[512];
    const uint64_t seq = c->sent[o.peer]++;
    name_of(tmp, sizeof tmp, c, c->rank, o.peer, seq, ".part");
    name_of(path, sizeof path, c, c->rank, o.peer, seq, "");
    int fd = open(tmp, O_CREAT | O_RDWR | O_TRUNC, 0600);
    if (fd < 0)
        return ncclSystemError;
    ncclResult_t rc = ncclSuccess;
    if (o.bytes) {
        if (ftruncate(fd, (off_t)o.bytes) != 0) {
            close(fd);
            return ncclSystemError;
        }
        void *map = mmap(NULL, o.bytes, PROT_READ | PROT_WRITE, MAP_SHARED,
                         fd, 0);
        if (map == MAP_FAILED) {
            close(fd);
            return ncclSystemError;
        }
        if (hipStreamSynchronize(o.stream) != hipSuccess ||
            hipMemcpy(map, o.buf, o.bytes, hipMemcpyDeviceToHost) != hipSuccess)
            rc = ncclUnhandledCudaError;
        munmap(map, o.bytes);
    }
    close(fd);
    if (rc == ncclSuccess && rename(tmp, path) != 0)
        rc = ncclSystemError;
    return rc;
}

ncclResult_t do_recv(const op &o)
{
    comm *c = o.c;
    char path[512];
    const uint64_t seq = c->received[o.peer]++;
    name_of(path, sizeof path, c, o.peer, c->rank, seq, "");
    const double deadline = now() + 120.;
    int fd;
    while ((fd = open(path, O_RDONLY)) < 0) {
        if (now() > deadline)
            return ncclRemoteError;
        usleep(200);
    }
    struct stat st;
    ncclResult_t rc = ncclSuccess;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size != o.bytes) {
        rc = ncclInvalidArgument;   /* the two sides disagree on the size */
    } else if (o.bytes) {
        void *map = mmap(NULL, o.bytes, PROT_READ, MAP_SHARED, fd, 0);
        if (map == MAP_FAILED) {
            rc = ncclSystemError;
        } else {
            if (hipStreamSynchronize(o.stream) != hipSuccess ||
                hipMemcpy(o.buf, map, o.bytes, hipMemcpyHostToDevice) !=
                    hipSuccess)
                rc = ncclUnhandledCudaError;
            munmap(map, o.bytes);
        }
    }
    close(fd);
    unlink(path);
    return rc;
}

ncclResult_t run(const op &o)
{
    return o.send ? do_send(o) : do_recv(o);
}

ncclResult_t submit(const op &o)
{
    if (!o.c || o.peer < 0 || o.peer >= o.c->nranks || o.peer == o.c->rank)
        return ncclInvalidArgument;
    if (group_depth > 0) {
        pending.push_back(o);
        return ncclSuccess;
    }
    return run(o);
}

} /* namespace */

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    if (!id)
        return ncclInvalidArgument;
    memset(id, 0, sizeof *id);
    timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "rt_shm_transport_%d_%lld_%ld",
             (int)getpid(), (long long)ts.tv_sec, (long)ts.tv_nsec);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id,
                              int rank)
{
    if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks)
        return ncclInvalidArgument;
    if (strncmp(id.internal, "rt_shm_transport_", 17) != 0)
        return ncclInvalidArgument;
    comm *c = new comm();
    c->nranks = nranks;
    c->rank = rank;
    snprintf(c->tag, sizeof c->tag, "%.90s", id.internal);
    *out = reinterpret_cast<ncclComm_t>(c);
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t handle)
{
    delete reinterpret_cast<comm *>(handle);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart(void)
{
    ++group_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd(void)
{
    if (group_depth <= 0)
        return ncclInvalidUsage;
    if (--group_depth > 0)
        return ncclSuccess;
    std::vector<op> ops;
    ops.swap(pending);
    ncclResult_t rc = ncclSuccess;
    for (int pass = 0; pass < 2; ++pass)        /* sends first, then receives */
        for (const op &o : ops)
            if (o.send == (pass == 0) && rc == ncclSuccess)
                rc = run(o);
    return rc;
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type,
                      int peer, ncclComm_t handle, hipStream_t stream)
{
    const size_t sz = type_size(type);
    if (!sz)
        return ncclInvalidArgument;
    return submit(op{true, const_cast<void *>(buf), count * sz, peer,
                     reinterpret_cast<comm *>(handle), stream});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer,
                      ncclComm_t handle, hipStream_t stream)
{
    const size_t sz = type_size(type);
    if (!sz)
        return ncclInvalidArgument;
    return submit(op{false, buf, count * sz, peer,
                     reinterpret_cast<comm *>(handle), stream});
}

ncclResult_t ncclCommCount(const ncclComm_t handle, int *count)
{
    if (!handle || !count)
        return ncclInvalidArgument;
    *count = reinterpret_cast<const comm *>(handle)->nranks;
    return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t handle, int *rank)
{
    if (!handle || !rank)
        return ncclInvalidArgument;
    *rank = reinterpret_cast<const comm *>(handle)->rank;
    return ncclSuccess;
}

ncclResult_t ncclGetVersion(int *version)
{
    if (!version)
        return ncclInvalidArgument;
    *version = -1; /* not RCCL: the stand-in */
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    switch (r) {
    case ncclSuccess: return "success (shm transport)";
    case ncclUnhandledCudaError: return "HIP error (shm transport)";
    case ncclSystemError: return "system error (shm transport)";
    case ncclInvalidArgument: return "invalid argument (shm transport)";
    case ncclInvalidUsage: return "invalid usage (shm transport)";
    case ncclRemoteError: return "peer did not send in time (shm transport)";
    default: return "error (shm transport)";
    }
}

} /* extern "C" */
