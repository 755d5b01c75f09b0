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
#
# Wire format: fixed 48-byte frame header + raw payload.  No pickle -- a job
# exchanges a 128-byte RCCL id, barrier tokens, a few floats / ints and (test
# mode only) float64 rows; nothing a peer sends is ever executed.
#
#   magic "RTHG" | kind u8 | ndim u8 | pad u16 | nbytes u64 | dims 4 x i64

_MAGIC = b"RTHG"
_HELLO = "rt-mi355-hostgroup-2"
_HEADER = struct.Struct("<4sBBHQ4q")
_NONE, _FLOAT, _INT, _BYTES, _F64, _I64, _BOOL, _STR = range(8)
_MAX_PAYLOAD = 1 << 30     # a frame: at most 1 GiB (rows of 10^8 rays: 0.8 GB)
_PROCESS_START = time.time()
_STALE_S = 600.     # a rendezvous file older than this process by more: stale


def _encode(obj):
    dims = [0, 0, 0, 0]
    if obj is None:
        kind, payload, ndim = _NONE, b"", 0
    elif isinstance(obj, (bool, np.bool_)):
        kind, payload, ndim = _BOOL, struct.pack("<q", int(obj)), 0
    elif isinstance(obj, (int, np.integer)):
        kind, payload, ndim = _INT, struct.pack("<q", int(obj)), 0
    elif isinstance(obj, (float, np.floating)):
        kind, payload, ndim = _FLOAT, struct.pack("<d", float(obj)), 0
    elif isinstance(obj, (bytes, bytearray, memoryview)):
        kind, payload, ndim = _BYTES, bytes(obj), 0
    elif isinstance(obj, str):
        kind, payload, ndim = _STR, obj.encode(), 0
    elif isinstance(obj, np.ndarray) and obj.ndim <= 4 and \
            obj.dtype in (np.float64, np.int64):
        kind = _F64 if obj.dtype == np.float64 else _I64
        payload, ndim = np.ascontiguousarray(obj).tobytes(), obj.ndim
        dims[:ndim] = obj.shape
    else:
        raise TypeError("host group carries None, bool, int, float, str, "
                        "bytes and float64 / int64 arrays, not %r"
                        % type(obj))
    if len(payload) > _MAX_PAYLOAD:
        # the receiver refuses such a header and stops reading while this
        # side would still be writing the payload: say it here instead
        raise ValueError("host group: a frame carries at most %d bytes, this "
                         "%s has %d -- send it in slices"
                         % (_MAX_PAYLOAD, type(obj).__name__, len(payload)))
    return _HEADER.pack(_MAGIC, kind, ndim, 0, len(payload), *dims) + payload


def _decode(kind, ndim, dims, payload):
    if kind == _NONE:
        return None
    if kind == _BOOL:
        return bool(struct.unpack("<q", payload)[0])
    if kind == _INT:
        return struct.unpack("<q", payload)[0]
    if kind == _FLOAT:
        return struct.unpack("<d", payload)[0]
    if kind == _BYTES:
        return payload
    if kind == _STR:
        return payload.decode()
    if kind in (_F64, _I64) and ndim <= 4:
        return np.frombuffer(payload, dtype=np.float64 if kind == _F64
                             else np.int64).reshape(dims[:ndim]).copy()
    raise ConnectionError("host group: unknown frame kind %d" % kind)


def _recv_exact(sock, n):
    chunks = []
    while n:
        chunk = sock.recv(min(n, 1 << 20))
        if not chunk:
            raise ConnectionError("peer closed the host-group connection")
        chunks.append(chunk)
        n -= len(chunk)
    return b"".join(chunks)


def _send(sock, obj):
    sock.sendall(_encode(obj))


def _recv(sock):
    magic, kind, ndim, _, nbytes, *dims = _HEADER.unpack(
        _recv_exact(sock, _HEADER.size))
    if magic != _MAGIC or nbytes > _MAX_PAYLOAD:
        raise ConnectionError("not a host-group frame")
    # the header is the peer's claim: checked BEFORE the payload is read
    if not 0 <= ndim <= 4 or any(d < 0 for d in dims[:ndim]):
        raise ConnectionError("host group: bad frame shape")
    if kind in (_F64, _I64):
        count = 1
        for d in dims[:ndim]:
            count *= d
        if count*8 != nbytes:
            raise ConnectionError("host group: frame of %d bytes for a "
                                  "shape of %d values" % (nbytes, count))
    elif kind in (_BOOL, _INT, _FLOAT) and nbytes != 8:
        raise ConnectionError("host group: bad scalar frame")
    return _decode(kind, ndim, dims, _recv_exact(sock, nbytes))


def _send_text(sock, text):
    data = text.encode()
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_text(sock, limit=4096):
    """A short text message of the handshake."""
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > limit:
        raise ConnectionError("not a host-group peer")
    return _recv_exact(sock, n).decode(errors="replace")


def rendezvous_path(env=None):
    """Where rank 0 publishes the port it listens on and the launch's token:
    ``RT_RDZV_FILE`` (set by :func:`spawn_workers`: a file in a private
    ``mkdtemp`` directory) wins; under an external launcher it is
    ``<tmp>/rt_mi355_rdzv_<uid>_<MASTER_PORT>_<run id>/rdzv`` -- a directory
    of mode 0700 that must belong to this user."""
    env = os.environ if env is None else env
    path = env.get("RT_RDZV_FILE")
    if path:
        # a file named inside a directory other users can write to (/tmp/x)
        # is kept in a private directory of this user next to it instead
        d = os.path.dirname(os.path.abspath(path)) or "."
        try:
            st = os.lstat(d)
            shared = st.st_uid != os.getuid() or bool(st.st_mode & 0o077)
        except OSError:
            shared = False      # does not exist yet: _private_dir makes it
        if shared:
            return os.path.join(d, "rt_rdzv_%d" % os.getuid(),
                                os.path.basename(path))
        return path
    tag = "%d_%s_%s" % (os.getuid(), env.get("MASTER_PORT", "0"),
                        env.get("TORCHELASTIC_RUN_ID", "none"))
    return os.path.join(tempfile.gettempdir(), "rt_mi355_rdzv_%s" % tag,
                        "rdzv")


def _private_dir(path):
    """Create (or accept) the directory of the rendezvous file: mode 0700,
    owned by this user, not a symlink -- another local user can neither read
    the token nor plant a file there."""
    d = os.path.dirname(path)
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or \
            st.st_mode & 0o077:
        raise PermissionError(
            "host group: rendezvous directory %s is not a private directory "
            "of this user" % d)
    return d


def _publish(path, text):
    """Write the rendezvous file: new inode (O_EXCL), mode 0600, then an
    atomic rename so readers never see half of it."""
    _private_dir(path)
    tmp = "%s.%d.tmp" % (path, os.getpid())
    try:
        os.unlink(tmp)
    except OSError:
        pass
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    with os.fdopen(fd, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read_published(path):
    """(port, token) from a rendezvous file that belongs to this user and
    nobody else can write."""
    fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0))
    with os.fdopen(fd) as f:
        st = os.fstat(f.fileno())
        if st.st_uid != os.getuid() or st.st_mode & 0o077:
            raise PermissionError("host group: %s is not private to this "
                                  "user" % path)
        # what an earlier launch left behind (same port, same run id) is not
        # this launch's rank 0: the ranks of one launch start within minutes
        if st.st_mtime < _PROCESS_START - _STALE_S:
            raise ValueError("stale rendezvous file")
        fields = f.read().split()
    return int(fields[0]), fields[1]


class HostGroup:
    """Star of TCP connections around rank 0: ``broadcast``, ``barrier``,
    ``allreduce_max``, ``gather``.  One node, a handful of ranks, messages of
    a few hundred bytes -- the device traffic goes through RCCL, not here.
    Every wait is bounded by ``timeout`` seconds and an expired one says
    which rank it was waiting for."""

    def __init__(self, world, rank, path=None, addr="127.0.0.1", timeout=120.):
        self.world, self.rank = int(world), int(rank)
        self.timeout = float(timeout)
        self.peers = []         # rank 0: socket of rank r at index r-1
        self.sock = None        # other ranks: connection to rank 0
        self._path = None
        if self.world == 1:
            return
        path = path or rendezvous_path()
        # spawn_workers hands every worker the launch's random token; under
        # an external launcher rank 0 draws one and the others read it from
        # the (private) rendezvous file
        token = os.environ.get("RT_RDZV_TOKEN", "")
        deadline = time.monotonic() + timeout
        if self.rank == 0:
            token = token or os.urandom(16).hex()
            server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            server.bind((addr, 0))
            server.listen(self.world)
            port = server.getsockname()[1]
            _publish(path, "%d %s\n" % (port, token))
            self._path = path
            slots = [None]*(self.world - 1)
            server.settimeout(1.)
            while any(s is None for s in slots):
                if time.monotonic() > deadline:
                    missing = [r + 1 for r, s in enumerate(slots) if s is None]
                    raise TimeoutError(
                        "host group: rank(s) %s of %d did not connect within "
                        "%.0f s" % (missing, self.world, timeout))
                try:
                    conn, _ = server.accept()
                except socket.timeout:
                    continue
                conn.settimeout(5.)     # a stranger must not hold rank 0 up
                try:
                    hello, theirs, peer = _recv_text(conn).split("\n")
                    peer = int(peer)
                except Exception:
                    conn.close()
                    continue
                conn.settimeout(timeout)
                if (hello != _HELLO or theirs != token or
                        not 1 <= peer < self.world or
                        slots[peer - 1] is not None):
                    conn.close()        # a stranger, or a stale launch
                    continue
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                slots[peer - 1] = conn
            server.close()
            self.peers = slots
            for conn in self.peers:
                _send_text(conn, _HELLO + "\nwelcome")
        else:
            while True:
                if time.monotonic() > deadline:
                    raise TimeoutError(
                        "host group: rank %d found no rank 0 at %s within "
                        "%.0f s" % (self.rank, path, timeout))
                try:
                    port, published = _read_published(path)
                    if token and published != token:
                        raise ValueError("stale rendezvous file")
                    sock = socket.create_connection((addr, port), timeout=5.)
                    sock.settimeout(timeout)
                    _send_text(sock, "%s\n%s\n%d" % (_HELLO, published,
                                                      self.rank))
                    if _recv_text(sock) != _HELLO + "\nwelcome":
                        raise ConnectionError("not the host group")
                except PermissionError:
                    raise
                except (OSError, ValueError, IndexError, EOFError,
                        struct.error):
                    time.sleep(.05)     # not published yet, or a stale file
                    continue
                sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                self.sock = sock
                break

    # -- collectives --------------------------------------------------------
    def _from(self, sock, peer, what):
        try:
            return _recv(sock)
        except socket.timeout:
            raise TimeoutError(
                "host group: rank %d waited %.0f s for rank %d in %s" % (
                    self.rank, self.timeout, peer, what)) from None

    def gather(self, obj, what="gather"):
        """List of every rank's ``obj`` on rank 0, None elsewhere."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            return [obj] + [self._from(conn, r + 1, what)
                            for r, conn in enumerate(self.peers)]
        _send(self.sock, obj)
        return None

    def broadcast(self, obj, src=0, what="broadcast"):
        """``obj`` of rank ``src`` on every rank."""
        if self.world == 1:
            return obj
        if src != 0:            # route through the hub
            box = self.gather(obj if self.rank == src else None, what)
            obj = box[src] if self.rank == 0 else None
        if self.rank == 0:
            for conn in self.peers:
                _send(conn, obj)
            return obj
        return self._from(self.sock, 0, what)

    def barrier(self, what="barrier"):
        self.broadcast(self.gather(None, what) is not None, what=what)

    def allreduce_max(self, value):
        box = self.gather(float(value), "allreduce_max")
        return self.broadcast(max(box) if self.rank == 0 else None,
                              what="allreduce_max")

    def allreduce_min(self, value):
        box = self.gather(float(value), "allreduce_min")
        return self.broadcast(min(box) if self.rank == 0 else None,
                              what="allreduce_min")

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
            if not os.environ.get("RT_RDZV_FILE"):  # the directory we made
                try:
                    os.rmdir(os.path.dirname(self._path))
                except OSError:
                    pass
            self._path = None


def init_engine_comm(engine, group, timeout=None):
    """Create the engine's RCCL communicator across ``group``: rank 0 draws
    the unique id (``rt_comm_unique_id``), everybody receives it over the
    host group and joins (``rt_comm_init``).  ``ncclCommInitRank`` blocks
    until every rank has joined and cannot be interrupted, so it runs on a
    helper thread: the barrier in front of it names a rank that never got
    this far, and a join that does not return within ``timeout`` seconds
    (default: the group's) raises instead of hanging the job."""
    import threading
    timeout = group.timeout if timeout is None else float(timeout)
    uid = engine.comm_unique_id() if group.rank == 0 else None
    uid = bytes(group.broadcast(uid, 0, what="the RCCL unique id"))
    group.barrier("the barrier before rt_comm_init")
    box = {}

    def join():
        try:
            engine.comm_init(uid, group.world, group.rank)
        except BaseException as err:        # re-raised on the caller's thread
            box["error"] = err
    worker = threading.Thread(target=join, daemon=True,
                              name="rt_comm_init")
    worker.start()
    worker.join(timeout)
    if worker.is_alive():
        raise TimeoutError(
            "rank %d: rt_comm_init (ncclCommInitRank, %d ranks) did not "
            "return within %.0f s although every rank reached the barrier "
            "in front of it: RCCL bootstrap / xGMI / IPC "
            "(HSA_ENABLE_IPC_MODE_LEGACY=0?)" % (group.rank, group.world,
                                                 timeout))
    if "error" in box:
        raise box["error"]
    group.barrier("the barrier after rt_comm_init")
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
    # a private directory (mode 0700) that exists for the life of the launch;
    # rank 0 creates the rendezvous file in it with O_EXCL, mode 0600
    private = tempfile.mkdtemp(prefix="rt_mi355_rdzv_")
    path = os.path.join(private, "rdzv")
    token = os.urandom(16).hex()
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
        for target, remove in ((path, os.unlink), (private, os.rmdir)):
            try:
                remove(target)
            except OSError:
                pass
    return code
