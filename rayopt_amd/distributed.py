"""Multi-GPU host logic: one process per GPU, contiguous ray shards.

The trace needs no communication (rays are independent, the surface table is
replicated).  The only exchange is the gather of the last-surface intercepts
to a root rank, done inside librt_mi355.so with RCCL send/recv over xGMI
(``rt_gather_final``).  This module holds the host-side bookkeeping around it:

* shard bounds and the layout of the gathered buffer;
* :class:`HostGroup` -- the few host-side collectives a job needs around the
  device work (hand out the 128-byte RCCL unique id, barrier, max of a
  timing), over plain TCP sockets on one node.  No PyTorch: the engine's
  device exchange is RCCL called from the C ABI, and the host side is a star
  of sockets around rank 0;
* :func:`spawn_workers` -- start one worker process per GPU from a plain
  ``python script.py --gpus N`` invocation (the environment a launcher such
  as ``python -m torch.distributed.run`` would have prepared -- ``RANK``,
  ``LOCAL_RANK``, ``WORLD_SIZE`` -- is honoured when it is there).

The reference has no parallelism at all; this is the
"ray batches shard trivially across the 8 GPUs of one node with an RCCL gather
over xGMI only for the final intercept arrays" of BASELINE.json's north star.
"""
import os
import pickle
import socket
import struct
import subprocess
import sys
import tempfile
import time

import numpy as np


def world_info(env=None):
    """(world_size, rank, local_rank) as a per-GPU launcher exports them."""
    env = os.environ if env is None else env
    return (int(env.get("WORLD_SIZE", "1")), int(env.get("RANK", "0")),
            int(env.get("LOCAL_RANK", env.get("RANK", "0"))))


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


def split_gathered(buf, counts, ncomp=3):
    """View the root's gathered buffer as (ncomp, total) and return the
    per-rank (count, ncomp) blocks in rank order."""
    counts = np.asarray(counts, dtype=np.int64)
    total = int(counts.sum())
    arr = np.asarray(buf).reshape(ncomp, total)
    offs = gather_offsets(counts)
    return [arr[:, o:o + c].T for o, c in zip(offs, counts)]


# --------------------------------------------------------------------------
# host-side process group over TCP (single node)
# --------------------------------------------------------------------------

_MAGIC = "rt-mi355-hostgroup-1"


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    chunks = []
    while n:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the host-group connection")
        chunks.append(chunk)
        n -= len(chunk)
    return b"".join(chunks)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    return pickle.loads(_recv_exact(sock, n))


def _send_text(sock, text):
    data = text.encode()
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_text(sock, limit=4096):
    """A short text message; nothing is unpickled before the peer has shown
    the launch's token."""
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError("not a host-group peer")
    return _recv_exact(sock, n).decode(errors="replace")


def rendezvous_path(env=None):
    """Where rank 0 publishes the port it listens on.  ``RT_RDZV_FILE`` (set
    by :func:`spawn_workers`) wins; under an external launcher the file is
    named after the launch's ``MASTER_PORT``, which is unique among the jobs
    running on a node at the same time."""
    env = os.environ if env is None else env
    path = env.get("RT_RDZV_FILE")
    if path:
        return path
    tag = "%s_%s" % (env.get("MASTER_PORT", "0"),
                     env.get("TORCHELASTIC_RUN_ID", "none"))
    return os.path.join(tempfile.gettempdir(), "rt_mi355_rdzv_%s" % tag)


class HostGroup:
    """Star of TCP connections around rank 0: ``broadcast``, ``barrier``,
    ``allreduce_max``, ``gather``.  One node, a handful of ranks, messages of
    a few hundred bytes -- the device traffic goes through RCCL, not here."""

    def __init__(self, world, rank, path=None, addr="127.0.0.1", timeout=120.):
        self.world, self.rank = int(world), int(rank)
        self.peers = []         # rank 0: socket of rank r at index r-1
        self.sock = None        # other ranks: connection to rank 0
        self._path = None
        if self.world == 1:
            return
        path = path or rendezvous_path()
        token = os.environ.get("RT_RDZV_TOKEN", "")
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            server.bind((addr, 0))
            server.listen(self.world)
            port = server.getsockname()[1]
            tmp = "%s.%d.tmp" % (path, os.getpid())
            with open(tmp, "w") as f:
                f.write("%d %s\n" % (port, token))
            os.replace(tmp, path)       # atomic: readers never see half
            self._path = path
            slots = [None]*(self.world - 1)
            server.settimeout(1.)
            while any(s is None for s in slots):
                if time.monotonic() > deadline:
                    raise TimeoutError(
                        "host group: %d of %d ranks connected" % (
                            1 + sum(s is not None for s in slots), self.world))
                try:
                    conn, _ = server.accept()
                except socket.timeout:
                    continue
                conn.settimeout(5.)     # a stranger must not hold rank 0 up
                try:
                    magic, theirs, peer = _recv_text(conn).split("\n")
                    peer = int(peer)
                except Exception:
                    conn.close()
                    continue
                conn.settimeout(timeout)
                if (magic != _MAGIC or theirs != token or
                        not 1 <= peer < self.world or
                        slots[peer - 1] is not None):
                    conn.close()        # a stranger, or a stale launch
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[peer - 1] = conn
            server.close()
            self.peers = slots
            for conn in self.peers:
                _send_text(conn, _MAGIC + "\nwelcome")
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError("host group: rank %d found no rank 0 "
                                       "at %s" % (self.rank, path))
                try:
                    with open(path) as f:
                        fields = f.read().split()
                    port = int(fields[0])
                    if (fields[1] if len(fields) > 1 else "") != token:
                        raise ValueError("stale rendezvous file")
                    sock = socket.create_connection((addr, port), timeout=5.)
                    sock.settimeout(timeout)
                    _send_text(sock, "%s\n%s\n%d" % (_MAGIC, token,
                                                      self.rank))
                    if _recv_text(sock) != _MAGIC + "\nwelcome":
                        raise ConnectionError("not the host group")
                except (OSError, ValueError, IndexError, EOFError,
                        struct.error):
                    time.sleep(.05)     # not published yet, or a stale file
                    continue
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.sock = sock
                break

    # -- collectives --------------------------------------------------------
    def gather(self, obj):
        """List of every rank's ``obj`` on rank 0, None elsewhere."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [_recv(conn) for conn in self.peers]
        _send(self.sock, obj)
        return None

    def broadcast(self, obj, src=0):
        """``obj`` of rank ``src`` on every rank."""
        if self.world == 1:
            return obj
        if src != 0:            # route through the hub
            box = self.gather(obj if self.rank == src else None)
            obj = box[src] if self.rank == 0 else None
        if self.rank == 0:
            for conn in self.peers:
                _send(conn, obj)
            return obj
        return _recv(self.sock)

    def barrier(self):
        self.broadcast(self.gather(None) is not None)

    def allreduce_max(self, value):
        box = self.gather(float(value))
        return self.broadcast(max(box) if self.rank == 0 else None)

    def allreduce_min(self, value):
        box = self.gather(float(value))
        return self.broadcast(min(box) if self.rank == 0 else None)

    def close(self):
        for conn in self.peers:
            conn.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None
        if self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
            self._path = None


def init_engine_comm(engine, group):
    """Create the engine's RCCL communicator across ``group``: rank 0 draws
    the unique id (``rt_comm_unique_id``), everybody receives it over the
    host group and joins (``rt_comm_init``)."""
    uid = engine.comm_unique_id() if group.rank == 0 else None
    uid = bytes(group.broadcast(uid, 0))
    engine.comm_init(uid, group.world, group.rank)
    return group.world, group.rank


def visible_devices():
    """Number of HIP devices this process can open (via the C ABI)."""
    import ctypes
    from . import _lib
    count = ctypes.c_int(0)
    _lib.load().rt_device_count(ctypes.byref(count))
    return count.value


def spawn_workers(world, argv=None, env=None, check_devices=True):
    """Run ``argv`` (default: this very command line) once per GPU with
    ``RANK / LOCAL_RANK / WORLD_SIZE`` set, a private rendezvous file and a
    launch token; wait for all of them.  stdout/stderr are inherited, so the
    one JSON line rank 0 prints is this process's output.  Returns the
    largest exit code; a rank that dies takes the others down."""
    world = int(world)
    if check_devices:
        have = visible_devices()
        if have < world:
            raise SystemExit("--gpus %d: %d devices needed, %d visible"
                             % (world, world, have))
    argv = [sys.executable] + sys.argv if argv is None else list(argv)
    base = dict(os.environ if env is None else env)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC for RCCL
    fd, path = tempfile.mkstemp(prefix="rt_mi355_rdzv_")
    os.close(fd)
    os.unlink(path)
    token = "%d-%d" % (os.getpid(), time.time_ns())
    procs = []
    for rank in range(world):
        procs.append(subprocess.Popen(argv, env=dict(
            base, RANK=str(rank), LOCAL_RANK=str(rank),
            WORLD_SIZE=str(world), RT_RDZV_FILE=path, RT_RDZV_TOKEN=token)))
    code = 0
    try:
        pending = set(range(world))
        while pending:
            for r in sorted(pending):
                rc = procs[r].poll()
                if rc is None:
                    continue
                pending.discard(r)
                if rc != 0:
                    code = max(code, rc if rc > 0 else 1)
                    for q in pending:       # do not leave ranks waiting
                        procs[q].terminate()
            time.sleep(.02)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink(path)
        except OSError:
            pass
    return code
