"""Multi-process path on CPU: world_size 2 (and 3), no GPU, no PyTorch in the
workers.  Covers the host logic of the N>1 path -- shard bounds, the host
group (rendezvous, broadcast of the RCCL unique id, barrier, max of the
timings), the self-spawning launcher of ``python bench.py --gpus N`` and the
layout of the gathered buffer.  The workers use a stand-in for the GPU
context (FakeEngine); the engine's own gather (rt_gather_final with
nranks > 1, rayopt_amd/csrc/rt_engine.hip) is covered on the device by
tests/test_gather_ranks_gpu.py over a stand-in transport; RCCL itself with
several ranks runs in ``bench.py --gpus N`` on a multi-GPU node only."""
import os
import socket
import struct
import time
import subprocess
import sys
import textwrap

import numpy as np
import pytest

from rayopt_amd import distributed as D
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
    assert shard_counts(10**8, 8).tolist() == [12_500_000]*8   # configs[4]


def test_split_gathered_layout():
    counts = np.array([3, 2, 4])
    total = counts.sum()
    buf = np.arange(3*total, dtype=float)      # [component][global ray]
    parts = split_gathered(buf, counts)
    assert [p.shape for p in parts] == [(3, 3), (2, 3), (4, 3)]
    assert parts[1][0].tolist() == [3., 3. + total, 3. + 2*total]


def test_single_rank_group_is_trivial():
    g = D.HostGroup(1, 0)
    assert g.broadcast(b"x") == b"x" and g.gather(3) == [3]
    assert g.allreduce_max(2.5) == 2.5
    g.barrier()
    g.close()


WORKER = textwrap.dedent("""
    import os, sys, time
    import numpy as np
    sys.path.insert(0, %r)
    assert "torch" not in sys.modules
    from rayopt_amd import distributed as D
    from rayopt_amd.bundles import disc_bundle

    class FakeEngine:
        # stands in for the GPU context: records what the host logic asks for
        def comm_unique_id(self):
            return bytes(range(128))
        def comm_init(self, uid, nranks, rank):
            self.args = (bytes(uid), nranks, rank)

    world, rank, local = D.world_info()
    if os.environ.get("RT_TEST_FAIL_RANK") == str(rank):
        sys.exit(7)
    group = D.HostGroup(world, rank)
    eng = FakeEngine()
    assert D.init_engine_comm(eng, group) == (world, rank)
    assert eng.args == (bytes(range(128)), world, rank)

    # barrier really waits for the slowest rank
    t0 = time.monotonic()
    if rank == world - 1:
        time.sleep(.3)
    group.barrier()
    assert time.monotonic() - t0 > .25
    assert group.allreduce_max(10. + rank) == 10. + world - 1
    assert group.allreduce_min(10. + rank) == 10.
    assert group.broadcast("from-%%d" %% rank, src=world - 1) == \\
        "from-%%d" %% (world - 1)

    # shard a global batch, "trace" it locally (identity stand-in), gather the
    # final rows the way the root lays them out, compare with the unsharded
    n = 1001
    y, u = disc_bundle(n, 3., 1., 0)
    lo, hi = D.shard_bounds(n, world)[rank]
    counts = D.shard_counts(n, world)
    box = group.gather(y[lo:hi])
    if rank == 0:
        buf = np.concatenate([b.T for b in box], axis=1).ravel()
        parts = D.split_gathered(buf, counts)
        assert np.array_equal(np.concatenate(parts), y)
    group.barrier()
    group.close()
    assert "torch" not in sys.modules
    sys.stdout.write("rank %%d ok\\n" %% rank)     # one write: ranks share a pipe
    sys.stdout.flush()
""")


@pytest.mark.parametrize("world", [2, 3])
def test_spawned_workers_host_group(tmp_path, world, capfd):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    rc = D.spawn_workers(world, [sys.executable, str(script)],
                         check_devices=False)
    out = capfd.readouterr().out
    assert rc == 0, out
    for rank in range(world):
        assert "rank %d ok" % rank in out


def test_a_dying_rank_takes_the_job_down(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, RT_TEST_FAIL_RANK="1")
    rc = D.spawn_workers(2, [sys.executable, str(script)], env=env,
                         check_devices=False)
    assert rc == 7


def test_under_a_per_gpu_launcher(tmp_path):
    """The environment `python -m torch.distributed.run` prepares (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_PORT): the workers find each other through
    the rendezvous file named after MASTER_PORT, without importing torch."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RT_RDZV_FILE", "RT_RDZV_TOKEN")}
    procs = []
    for rank in range(2):
        e = dict(env, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank),
                 MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=e,
                                      stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, out
        assert "rank %d ok" % rank in out


def test_bench_needs_as_many_devices_as_gpus():
    """`python bench.py --gpus 2` as typed: no launcher message, a clear
    statement of what is missing (this container has no GPU at all)."""
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"], cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120)
    assert res.returncode != 0
    assert res.stdout.strip() == ""
    have = D.visible_devices()
    if have < 2:
        assert "2 devices needed, %d visible" % have in res.stderr


def test_under_torch_distributed_run(tmp_path):
    """The contract's launch line for N > 1 -- `python -m
    torch.distributed.run --nproc-per-node 2 ...` -- around the torch-free
    workers: they take RANK / LOCAL_RANK / WORLD_SIZE from it and meet
    through the MASTER_PORT-named rendezvous file (the launcher's own store
    occupies that port)."""
    pytest.importorskip("torch")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items()
           if k not in ("RT_RDZV_FILE", "RT_RDZV_TOKEN")}
    res = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
         "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port), str(script)], env=env, text=True,
        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert res.returncode == 0, res.stdout[-2000:]
    assert "rank 0 ok" in res.stdout and "rank 1 ok" in res.stdout


def test_frames_round_trip_and_nothing_is_unpickled():
    """The host group's wire format is a fixed header + raw bytes: every
    payload kind a job sends survives, anything else is refused on both
    ends, and a pickle sent by a stranger is never loaded."""
    import pickle
    a, b = socket.socketpair()
    try:
        for obj in (None, True, 7, -2.5, "text", bytes(range(128)),
                    np.arange(12.).reshape(3, 4),
                    np.arange(5, dtype=np.int64)):
            D._send(a, obj)
            got = D._recv(b)
            if isinstance(obj, np.ndarray):
                assert got.dtype == obj.dtype and np.array_equal(got, obj)
            else:
                assert got == obj and type(got) is type(obj)
        with pytest.raises(TypeError):
            D._send(a, {"a": 1})
        evil = pickle.dumps(os.system)
        a.sendall(struct.pack("<Q", len(evil)) + evil + b"\0"*64)
        with pytest.raises(ConnectionError):
            D._recv(b)
    finally:
        a.close()
        b.close()
    assert "pickle" not in open(D.__file__).read().replace(
        "No pickle", "")


def test_expired_wait_names_the_missing_rank(tmp_path):
    """Every host-group wait is bounded; the error says who did not come."""
    import threading
    path = str(tmp_path / "private" / "rdzv")
    os.mkdir(os.path.dirname(path), 0o700)
    errors = {}

    def rank1():
        g = D.HostGroup(2, 1, path=path, timeout=20.)
        time.sleep(3.)          # never takes part in the barrier
        g.close()
    t = threading.Thread(target=rank1)
    t.start()
    g = D.HostGroup(2, 0, path=path, timeout=1.)
    with pytest.raises(TimeoutError, match="for rank 1 in my barrier"):
        g.barrier("my barrier")
    g.close()
    t.join()
    # a rank that never connects is named as well
    with pytest.raises(TimeoutError, match=r"rank\(s\) \[1, 2\] of 3"):
        D.HostGroup(3, 0, path=path, timeout=1.)


def test_rendezvous_is_private_to_the_user(tmp_path):
    path = str(tmp_path / "d" / "rdzv")
    os.mkdir(os.path.dirname(path), 0o700)
    D._publish(path, "1234 token\n")
    assert os.stat(path).st_mode & 0o777 == 0o600
    assert D._read_published(path) == (1234, "token")
    os.chmod(path, 0o644)
    with pytest.raises(PermissionError):
        D._read_published(path)
    os.chmod(os.path.dirname(path), 0o755)
    with pytest.raises(PermissionError):
        D._publish(path, "1 t\n")
    # the path an external launcher leads to carries the uid and lies in a
    # directory of its own
    p = D.rendezvous_path({"MASTER_PORT": "29500"})
    assert ("_%d_29500_" % os.getuid()) in p and p.endswith("/rdzv")


def test_a_frame_header_is_checked_before_its_payload_is_read():
    """ndim, dims and the byte count of a frame are the peer's claims: a frame
    whose shape does not add up to its payload is refused as a broken
    connection (not a ValueError out of reshape), without waiting for bytes
    that will never come."""
    a, b = socket.socketpair()
    b.settimeout(2.)
    try:
        good = D._encode(np.arange(6.).reshape(2, 3))
        magic, kind, ndim, pad, nbytes, *dims = D._HEADER.unpack(
            good[:D._HEADER.size])
        for bad in (dict(ndim=7), dict(dims=[2, 4, 0, 0]), dict(nbytes=40),
                    dict(dims=[-2, -3, 0, 0])):
            hdr = D._HEADER.pack(magic, kind, bad.get("ndim", ndim), pad,
                                 bad.get("nbytes", nbytes),
                                 *bad.get("dims", dims))
            a.sendall(hdr)          # no payload follows
            with pytest.raises(ConnectionError):
                D._recv(b)
        scalar = D._encode(3)
        hdr = bytearray(scalar[:D._HEADER.size])
        fields = list(D._HEADER.unpack(bytes(hdr)))
        fields[4] = 16
        a.sendall(D._HEADER.pack(*fields))
        with pytest.raises(ConnectionError):
            D._recv(b)
    finally:
        a.close()
        b.close()


def test_an_oversized_frame_is_refused_by_the_sender(monkeypatch):
    """The receiver drops a header that claims more than _MAX_PAYLOAD bytes
    and stops reading; a sender that wrote such a frame anyway would block in
    sendall with the connection out of step.  The sender says so
    before a byte leaves."""
    monkeypatch.setattr(D, "_MAX_PAYLOAD", 1 << 10)
    assert D._decode(*_parts(D._encode(np.zeros(128)))).shape == (128,)
    with pytest.raises(ValueError, match="at most 1024 bytes"):
        D._encode(np.zeros(129))
    with pytest.raises(ValueError, match="slices"):
        D._encode(b"x"*1025)


def _parts(frame):
    magic, kind, ndim, _, nbytes, *dims = D._HEADER.unpack(
        frame[:D._HEADER.size])
    return kind, ndim, dims, frame[D._HEADER.size:]


def test_rendezvous_file_in_a_shared_directory_and_stale_files(tmp_path):
    """RT_RDZV_FILE=/tmp/x style paths (a directory others can write to) are
    kept in a private directory of this user next to it; a file an earlier
    launch left behind is not this launch's rank 0."""
    shared = tmp_path / "shared"
    shared.mkdir()
    os.chmod(shared, 0o777)
    p = D.rendezvous_path({"RT_RDZV_FILE": str(shared / "x")})
    assert p == str(shared / ("rt_rdzv_%d" % os.getuid()) / "x")
    D._publish(p, "4321 tok\n")
    assert os.stat(os.path.dirname(p)).st_mode & 0o777 == 0o700
    assert D._read_published(p) == (4321, "tok")
    # a private directory is used as given
    own = tmp_path / "own"
    own.mkdir(mode=0o700)
    assert D.rendezvous_path({"RT_RDZV_FILE": str(own / "x")}) == \
        str(own / "x")
    # stale: older than this process by more than the allowance
    old = time.time() - D._STALE_S - 3600
    os.utime(p, (old, old))
    with pytest.raises(ValueError, match="stale"):
        D._read_published(p)
