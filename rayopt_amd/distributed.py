"""Multi-GPU host logic: one process per GPU, contiguous ray shards.

The trace needs no communication (rays are independent, the surface table is
replicated).  The only exchange is the gather of the last-surface intercepts
to a root rank, done inside librt_mi355.so with RCCL send/recv over xGMI
(``rt_gather_final``).  This module holds the host-side bookkeeping around it:
shard bounds, distribution of the RCCL unique id through whatever process
group the launcher provides (``torch.distributed`` -- nccl on GPUs, gloo in
the CPU tests), and the layout of the gathered buffer.
"""
import os

import numpy as np


def world_info(env=None):
    """(world_size, rank, local_rank) as torch.distributed.run exports them."""
    env = os.environ if env is None else env
    return (int(env.get("WORLD_SIZE", "1")), int(env.get("RANK", "0")),
            int(env.get("LOCAL_RANK", "0")))


def shard_bounds(nrays, world):
    """Contiguous, balanced [lo, hi) per rank; the first ``nrays % world``
    ranks hold one extra ray.  Concatenating the shards in rank order
    restores the global ray order."""
    base, extra = divmod(int(nrays), int(world))
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < extra else 0)
        bounds.append((lo, hi))
        lo = hi
    return bounds


def shard_counts(nrays, world):
    return np.array([hi - lo for lo, hi in shard_bounds(nrays, world)],
                    dtype=np.int64)


def gather_offsets(counts):
    """Offset of each rank's rays inside the gathered [component][ray]
    buffer on the root (same arithmetic as rt_gather_final)."""
    counts = np.asarray(counts, dtype=np.int64)
    return np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int64)


def broadcast_bytes(dist, payload, src=0):
    """Hand ``payload`` (bytes on ``src``, ignored elsewhere) to every rank
    of the default process group."""
    box = [payload if dist.get_rank() == src else None]
    dist.broadcast_object_list(box, src=src)
    return bytes(box[0])


def init_engine_comm(engine, dist):
    """Create the engine's RCCL communicator across the process group: rank 0
    draws the unique id, everybody receives it out of band."""
    world, rank = dist.get_world_size(), dist.get_rank()
    uid = engine.comm_unique_id() if rank == 0 else None
    uid = broadcast_bytes(dist, uid, 0)
    engine.comm_init(uid, world, rank)
    return world, rank


def split_gathered(buf, counts, ncomp=3):
    """View the root's gathered buffer as (ncomp, total) and return the
    per-rank (count, ncomp) blocks in rank order."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    arr = np.asarray(buf).reshape(ncomp, total)
    offs = gather_offsets(counts)
    return [arr[:, o:o + c].T for o, c in zip(offs, counts)]
