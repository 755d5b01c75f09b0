"""Multi-process path on CPU: world_size 2, gloo.  Covers the host logic of
the N>1 path (shard bounds, unique-id distribution, gathered-buffer layout);
the RCCL transfer itself needs GPUs and is exercised by bench.py --gpus N."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

from rayopt_amd.distributed import (shard_bounds, shard_counts,
                                    gather_offsets, split_gathered)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    for n in (1, 7, 64, 10**7, 10**8 + 3):
        for w in (1, 2, 3, 8):
            b = shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            c = shard_counts(n, w)
            assert c.sum() == n and c.max() - c.min() <= 1
            assert list(gather_offsets(c)) == [lo for lo, _ in b]


def test_split_gathered_layout():
    counts = np.array([3, 2, 4])
    total = counts.sum()
    buf = np.arange(3*total, dtype=float)      # [component][global ray]
    parts = split_gathered(buf, counts)
    assert [p.shape for p in parts] == [(3, 3), (2, 3), (4, 3)]
    assert parts[1][0].tolist() == [3., 3. + total, 3. + 2*total]


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    import torch.distributed as dist
    sys.path.insert(0, %r)
    from rayopt_amd import distributed as D
    from rayopt_amd.bundles import disc_bundle

    class FakeEngine:
        # stands in for the GPU context: records what the host logic asks for
        def comm_unique_id(self):
            return bytes(range(128))
        def comm_init(self, uid, nranks, rank):
            self.args = (bytes(uid), nranks, rank)

    dist.init_process_group("gloo")
    world, rank, local = D.world_info()
    assert (world, rank) == (dist.get_world_size(), dist.get_rank())
    eng = FakeEngine()
    assert D.init_engine_comm(eng, dist) == (world, rank)
    assert eng.args == (bytes(range(128)), world, rank)

    # shard a global batch, "trace" it locally (identity stand-in), gather the
    # final rows the way the root lays them out, compare with the unsharded
    n = 1001
    y, u = disc_bundle(n, 3., 1., 0)
    lo, hi = D.shard_bounds(n, world)[rank]
    mine = y[lo:hi]
    counts = D.shard_counts(n, world)
    box = [None]*world
    dist.all_gather_object(box, mine)
    if rank == 0:
        buf = np.concatenate([b.T for b in box], axis=1).ravel()
        parts = D.split_gathered(buf, counts)
        assert np.array_equal(np.concatenate(parts), y)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2",
                   LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env,
                                      stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=300)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out
