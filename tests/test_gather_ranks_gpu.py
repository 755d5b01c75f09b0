"""rt_gather_final with nranks > 1 on ONE device.

RCCL refuses two ranks on one GPU and the boxes this project is built on have
one, so the engine's library loader is pointed (RT_TRANSPORT_LIBRARY) at the
shared-memory stand-in of tests/stubs/shm_transport.cpp.  Everything on the
engine's side is the real thing: the workers are separate processes with one
rt_ctx each, they trace their own shard on the device, rt_gather_final
snapshots the row into its double-buffered staging area and issues the grouped
ncclSend / ncclRecv calls with the offsets and counts it computes -- only the
bytes move through /dev/shm instead of xGMI.  Not covered: RCCL itself.
"""
import os
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from rayopt_amd import distributed as D

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB_SRC = os.path.join(ROOT, "tests", "stubs", "shm_transport.cpp")
STUB_LIB = os.path.join(ROOT, "tests", "stubs", "librt_shm_transport.so")


def build_stub():
    if (not os.path.exists(STUB_LIB) or
            os.path.getmtime(STUB_LIB) < os.path.getmtime(STUB_SRC)):
        subprocess.check_call(["hipcc", "-O2", "-std=c++17", "-fPIC",
                               "-shared", "-o", STUB_LIB, STUB_SRC])
    return STUB_LIB


WORKER = textwrap.dedent("""
    import os, sys
    import numpy as np
    sys.path.insert(0, %(root)r)
    assert "torch" not in sys.modules
    import rayopt_amd as ra
    from rayopt_amd import distributed as D
    from rayopt_amd._lib import RT_Y, RT_U, RT_T
    from rayopt_amd.bundles import disc_bundle

    world, rank, _ = D.world_info()
    group = D.HostGroup(world, rank)
    counts = np.array(%(counts)r, dtype=np.int64)
    root = %(root_rank)d
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    L = len(system)
    n = int(counts[rank])
    g = ra.GeometricTrace(system, device=0)       # every rank on device 0
    eng = g.engine
    D.init_engine_comm(eng, group)
    total = int(counts.sum())
    d3 = eng.scratch(total*3*8) if rank == root else 0
    bad = []
    # three rounds: the staging buffers alternate, the rays change every
    # round, and a T row (one component) goes between the Y rows
    for rnd, field in enumerate((0., 9., 14.)):
        y, u = disc_bundle(n, 12., field, 100*rnd + rank)
        g.rays_given(y, u)
        g.propagate(clip=True)
        for which, nc, rows in ((RT_Y, 3, g.y), (RT_T, 1, g.t),
                                (RT_U, 3, g.u)):
            surf = L - 1 if which != RT_U else L - 2
            eng.gather_final(which, surf, counts, root, d3)
            eng.comm_sync()
            mine = np.asarray(rows[surf]).reshape(n, nc)
            box = group.gather(mine)                # the same rows, by TCP
            have = group.gather(
                eng.copy_to_host(d3, total*nc*8) if rank == root else None)
            if rank == 0:
                parts = D.split_gathered(have[root], counts, nc)
                for r in range(world):
                    if not np.array_equal(parts[r], box[r], equal_nan=True):
                        bad.append((rnd, which, r))
                if not np.isfinite(have[root]).any():
                    bad.append((rnd, which, 'nothing finite'))
    # the row is snapshotted when the gather is issued: a trace of other rays
    # queued right behind it must not change what arrives
    want = group.gather(np.asarray(g.y[L - 1]))
    y, u = disc_bundle(n, 12., 3., 999 + rank)
    eng.gather_final(RT_Y, L - 1, counts, root, d3)
    g.rays_given(y, u)
    g.propagate(clip=True)
    eng.comm_sync()
    have = group.gather(
        eng.copy_to_host(d3, total*3*8) if rank == root else None)
    if rank == 0:
        for r, part in enumerate(D.split_gathered(have[root], counts, 3)):
            if not np.array_equal(part, want[r], equal_nan=True):
                bad.append(("snapshot", r))
    group.barrier()
    if rank == 0:
        assert not bad, bad
        print("gather ok", world, counts.tolist(), flush=True)
    group.barrier()
    eng.comm_destroy()
    group.close()
""")


def run(world, counts, root_rank=0, timeout=600):
    lib = build_stub()
    script = WORKER % {"root": ROOT, "counts": list(counts),
                       "root_rank": root_rank}
    env = dict(os.environ, RT_TRANSPORT_LIBRARY=lib)
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".py", delete=False) as f:
        f.write(script)
        path = f.name
    try:
        res = subprocess.run(
            [sys.executable, "-c",
             "import sys; sys.path.insert(0, %r); "
             "from rayopt_amd import distributed as D; "
             "raise SystemExit(D.spawn_workers(%d, argv=[sys.executable, %r], "
             "check_devices=False))" % (ROOT, world, path)],
            text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
            timeout=timeout, env=env, cwd=ROOT)
    finally:
        os.unlink(path)
    assert res.returncode == 0, res.stderr[-3000:]
    assert "gather ok %d" % world in res.stdout, res.stdout + res.stderr[-2000:]


def test_two_ranks_gather_uneven_shards():
    """counts that are not multiples of 64 and differ between the ranks:
    the gathered buffer is [component][global ray] with rank r's rays at
    offset sum(counts[:r]) -- compared with what each rank holds, all rays."""
    run(2, [100_003, 70_001])


def test_three_ranks_gather_root_in_the_middle():
    run(3, [4097, 65, 20_000], root_rank=1)


def test_missing_transport_library_is_loud():
    env = dict(os.environ, RT_TRANSPORT_LIBRARY="/nonexistent/libnope.so")
    res = subprocess.run(
        [sys.executable, "-c",
         "import sys; sys.path.insert(0, %r); "
         "from rayopt_amd.engine import Engine; "
         "Engine(0).comm_unique_id()" % ROOT],
        env=env, text=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert res.returncode != 0 and "libnope.so" in res.stderr


@pytest.mark.parametrize("every_step", [False, True])
def test_bench_two_ranks_one_device_real_gather(every_step):
    """`python bench.py --gpus 2` end to end on the one device: self-spawned
    ranks, host group, communicator, the job's gather inside the timed
    region (or in every step, the snapshot of step k+1 queued behind the
    transfer of step k), every gathered shard checked on rank 0; the line
    says test_mode."""
    import json
    env = dict(os.environ, RT_BENCH_SHARE_DEVICE="1",
               RT_TRANSPORT_LIBRARY=build_stub())
    out = subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
         "--total-rays", "400001", "--steps", "3", "--warmup", "1",
         "--settle", "0"] + (["--gather-every-step"] if every_step else []),
        text=True, cwd=ROOT, env=env,
        stderr=subprocess.DEVNULL, timeout=600)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and "stand-in" in d["test_mode"]
    assert d["config"]["total_rays"] == 400001
    assert d["gather_ms"] > 0 and len(d["kernel_ms_per_rank"]) == 2
