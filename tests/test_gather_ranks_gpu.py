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
    # the chunked form: the step is traced in pieces and the gather of piece
    # k is issued before piece k+1 is traced (rt_trace_chunk /
    # rt_gather_chunk); all pieces together = the unchunked trace + gather,
    # also where a rank's shard is smaller than one piece
    for chunks in (1, 3, 4, 7):
        y, u = disc_bundle(n, 12., 5., 4000 + 10*chunks + rank)
        g.rays_given(y, u)
        g.propagate(clip=True)
        want_rows = [np.array(np.asarray(r[:])) for r in (g.y, g.u, g.i, g.t)]
        g.rays_given(y, u)
        g.propagate(clip=True, chunks=chunks,
                    after_chunk=lambda k, q: eng.gather_chunk(
                        RT_Y, L - 1, counts, root, d3, k, q))
        eng.comm_sync()
        for a, rows in zip(want_rows, (g.y, g.u, g.i, g.t)):
            if not np.array_equal(a, np.asarray(rows[:]), equal_nan=True):
                bad.append(("chunked trace differs", chunks, rank))
        total_ms, exposed_ms = eng.gather_ms()
        if not (total_ms >= 0. and exposed_ms >= 0.):
            bad.append(("gather_ms", total_ms, exposed_ms))
        box = group.gather(want_rows[0][L - 1])
        have = group.gather(
            eng.copy_to_host(d3, total*3*8) if rank == root else None)
        flags = group.gather(len(bad))
        if rank == 0:
            for r, part in enumerate(D.split_gathered(have[root], counts, 3)):
                if not np.array_equal(part, box[r], equal_nan=True):
                    bad.append(("chunked gather", chunks, r))
            if any(flags[1:]):
                bad.append(("a worker saw a difference", flags))
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


def test_eight_ranks_uneven_shards_root_not_zero():
    """BASELINE configs[4]'s shape -- eight ranks -- at a reduced ray count:
    uneven shards (one smaller than a 256-ray workgroup, one smaller than a
    wavefront), the root in the middle of the node."""
    run(8, [30_001, 255, 12_800, 63, 20_000, 4097, 1, 9_999], root_rank=5,
        timeout=900)


def test_chunk_bounds_tile_the_batch():
    """rt_chunk_bounds: whole 256-ray workgroups, contiguous, covering
    [0, n) exactly once for every n and chunk count."""
    from rayopt_amd.engine import Engine
    eng = Engine(0)
    for n in (1, 63, 256, 257, 4097, 10**7, 12_500_000):
        for q in (1, 2, 3, 4, 7, 8, 64):
            edge = 0
            for k in range(q):
                lo, hi = eng.chunk_bounds(n, k, q)
                assert lo == edge and lo <= hi <= n and lo % 256 == 0 or \
                    lo == n
                edge = hi
            assert edge == n


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
    if not every_step:
        assert d["gather_chunks"] == 4 and d["gather_exposed_ms"] >= 0.
        assert d["plain_loop_ms_per_step"] > 0 and "exchange" in d["note"]


def test_bench_eight_ranks_one_device_configs4_shape():
    """`python bench.py --gpus 8 --total-rays N` -- the literal configs[4]
    command at a reduced N -- over the stand-in transport: eight self-spawned
    ranks, uneven shards, the chunked last step with its pipelined gather,
    every gathered shard checked on rank 0."""
    import json
    env = dict(os.environ, RT_BENCH_SHARE_DEVICE="1",
               RT_TRANSPORT_LIBRARY=build_stub())
    out = subprocess.check_output(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8",
         "--total-rays", "800003", "--steps", "3", "--warmup", "1",
         "--settle", "0"], text=True, cwd=ROOT, env=env,
        stderr=subprocess.DEVNULL, timeout=900)
    lines = [ln for ln in out.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and "stand-in" in d["test_mode"]
    assert d["config"]["total_rays"] == 800003
    assert len(d["kernel_ms_per_rank"]) == 8 and d["gather_chunks"] == 4
