"""Python face of one ``rt_ctx``: a device context of the HIP engine.

All ray arithmetic happens in librt_mi355.so on the GPU; this module only
moves arrays across the ctypes boundary.  Nothing here computes on the CPU
and nothing falls back to the CPU: without the library or without a GPU every
entry point raises :class:`EngineError`.
"""
import ctypes
import os
import sys

import numpy as np

from . import _lib
from ._lib import EngineError, RT_Y, RT_U, RT_I, RT_T, LAYOUT_AOS, LAYOUT_SOA
from .pack import pack_system, resolve_range


def _default_device():
    for key in ("RT_DEVICE", "LOCAL_RANK"):
        if key in os.environ:
            return int(os.environ[key])
    return 0


class Engine:
    """Owns one device context (stream, events, result arrays in HBM)."""

    def __init__(self, device=None, lib_path=None):
        self.lib = _lib.load(lib_path)
        self.device = _default_device() if device is None else int(device)
        ctx = ctypes.c_void_p()
        rc = self.lib.rt_create(self.device, ctypes.byref(ctx))
        if rc != 0:
            raise EngineError("rt_create(device=%d) failed (%d): %s" % (
                self.device, rc,
                (self.lib.rt_last_error(None) or b"").decode()))
        self.ctx = ctx

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.rt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise EngineError("%s failed (%d): %s" % (
                what, rc, (self.lib.rt_last_error(self.ctx) or b"").decode()))

    # -- system / rays ----------------------------------------------------
    def upload_system(self, table):
        """``table``: (L,) one surface table, or (G,L) one per ray group
        (wavelength)."""
        table = np.ascontiguousarray(table, dtype=_lib.SURFACE_DTYPE)
        if table.ndim == 1:
            table = table[None]
        groups, nsurf = table.shape
        self._check(self.lib.rt_upload_system_groups(
            self.ctx, table.ctypes.data, nsurf, groups), "rt_upload_system")
        self.nsurf = nsurf
        self.ngroups = groups

    def reserve(self, nrays):
        self._check(self.lib.rt_reserve(self.ctx, int(nrays)), "rt_reserve")

    @property
    def nrays(self):
        return int(self.lib.rt_nrays(self.ctx))

    @property
    def ld(self):
        return int(self.lib.rt_ld(self.ctx))

    def blocks(self):
        """(blocks, rays per block = row pitch, doubles between blocks): how
        a large batch is cut up on the device (rt_blocks); (1, ld, 0) for
        the plain layout."""
        info = (ctypes.c_int64*3)()
        self._check(self.lib.rt_blocks(self.ctx, info), "rt_blocks")
        return int(info[0]), int(info[1]), int(info[2])

    def set_rays(self, y, u):
        """Seed row 0 from host arrays, (N,3) ray-major or (3,N) via
        ``set_rays_soa``."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if y.shape != u.shape or y.ndim != 2 or y.shape[1] != 3:
            raise ValueError("y and u must both be (N,3)")
        self._check(self.lib.rt_set_rays(
            self.ctx, y.ctypes.data, u.ctypes.data, y.shape[0], LAYOUT_AOS),
            "rt_set_rays")

    def set_rays_repeat(self, y, u, copies):
        """The same (P,3) rays ``copies`` times in a row (one copy per ray
        group); only P rays cross PCIe."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if y.shape != u.shape or y.ndim != 2 or y.shape[1] != 3:
            raise ValueError("y and u must both be (P,3)")
        self._check(self.lib.rt_set_rays_repeat(
            self.ctx, y.ctypes.data, u.ctypes.data, y.shape[0], int(copies),
            LAYOUT_AOS), "rt_set_rays_repeat")

    def set_rays_soa(self, y, u):
        y = np.ascontiguousarray(y, dtype=np.float64)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if y.shape != u.shape or y.ndim != 2 or y.shape[0] != 3:
            raise ValueError("y and u must both be (3,N)")
        self._check(self.lib.rt_set_rays(
            self.ctx, y.ctypes.data, u.ctypes.data, y.shape[1], LAYOUT_SOA),
            "rt_set_rays")

    def set_rays_device(self, d_y, d_u, n, layout=LAYOUT_SOA):
        self._check(self.lib.rt_set_rays_device(
            self.ctx, ctypes.c_void_p(d_y), ctypes.c_void_p(d_u), int(n),
            layout), "rt_set_rays_device")

    def generate_rays(self, fields, pupil_xy):
        """Seed row 0 with len(fields) x len(pupil_xy) rays built on the
        device (rt_generate_rays); returns the ray count."""
        fields = np.ascontiguousarray(fields, dtype=_lib.FIELD_DTYPE)
        pupil = np.ascontiguousarray(pupil_xy, dtype=np.float64)
        if pupil.ndim != 2 or pupil.shape[1] != 2:
            raise ValueError("pupil coordinates must be (P,2)")
        self._check(self.lib.rt_generate_rays(
            self.ctx, fields.ctypes.data, len(fields), pupil.ctypes.data,
            pupil.shape[0]), "rt_generate_rays")
        return len(fields)*pupil.shape[0]

    def aim_pupil(self, seeds, args):
        """System.pupil for every field on the device (rt_aim_pupil):
        returns z (F,), a (F,2,2), status (F,) int32."""
        seeds = np.ascontiguousarray(seeds, dtype=_lib.AIM_SEED_DTYPE)
        args = np.ascontiguousarray(args, dtype=_lib.AIM_ARGS_DTYPE)
        nf = len(seeds)
        z, a = np.empty(nf), np.empty((nf, 2, 2))
        status = np.empty(nf, dtype=np.int32)
        self._check(self.lib.rt_aim_pupil(
            self.ctx, seeds.ctypes.data, nf, args.ctypes.data, z.ctypes.data,
            a.ctypes.data, status.ctypes.data), "rt_aim_pupil")
        return z, a, status

    def upload_row(self, which, surf, src_soa):
        src = np.ascontiguousarray(src_soa, dtype=np.float64)
        self._check(self.lib.rt_upload_row(self.ctx, which, surf,
                                           src.ctypes.data), "rt_upload_row")

    # -- trace ------------------------------------------------------------
    def trace(self, start=1, stop=0, clip=False):
        self._check(self.lib.rt_trace(self.ctx, int(start), int(stop),
                                      1 if clip else 0), "rt_trace")

    def newton_census(self, clip=False):
        """How the asphere iteration's per-wavefront trip count fits this
        batch (rt_newton_census): dict with the lane slots spent, the
        iterates the rays needed, their ratio (``lane_utilisation``), mean
        trips per wavefront solve and mean iterates per lane slot entered."""
        out = (ctypes.c_uint64*4)()
        self._check(self.lib.rt_newton_census(self.ctx, 1 if clip else 0,
                                              out), "rt_newton_census")
        slots, its, trips, solves = (int(v) for v in out)
        return {"lane_slots": slots, "iterates": its, "wave_trips": trips,
                "wave_solves": solves,
                "lane_utilisation": its/slots if slots else None,
                "trips_per_wave_solve": trips/solves if solves else None}

    def set_keep_rows(self, keep):
        """Rows propagate() stores: boolean sequence per element, or None
        for all rows (the reference behaviour)."""
        if keep is None:
            self._check(self.lib.rt_set_keep_rows(self.ctx, None, 0),
                        "rt_set_keep_rows")
            return
        keep = np.ascontiguousarray(keep, dtype=np.uint8)
        self._check(self.lib.rt_set_keep_rows(self.ctx, keep.ctypes.data,
                                              len(keep)), "rt_set_keep_rows")

    def sync(self):
        self._check(self.lib.rt_sync(self.ctx), "rt_sync")

    def kernel_ms(self):
        ms = ctypes.c_double()
        self._check(self.lib.rt_kernel_ms(self.ctx, ctypes.byref(ms)),
                    "rt_kernel_ms")
        return ms.value

    def event_record(self, slot):
        self._check(self.lib.rt_event_record(self.ctx, slot),
                    "rt_event_record")

    def event_elapsed(self, a, b):
        ms = ctypes.c_double()
        self._check(self.lib.rt_event_elapsed(self.ctx, a, b,
                                              ctypes.byref(ms)),
                    "rt_event_elapsed")
        return ms.value

    def set_option(self, key, value):
        self._check(self.lib.rt_set_option(self.ctx, key.encode(), int(value)),
                    "rt_set_option(%s)" % key)

    def trace_chunk(self, start, stop, clip, chunk, nchunks):
        """The trace for chunk ``chunk`` of ``nchunks`` of the rays
        (rt_trace_chunk); all chunks together = :meth:`trace`."""
        self._check(self.lib.rt_trace_chunk(
            self.ctx, int(start), int(stop), 1 if clip else 0, int(chunk),
            int(nchunks)), "rt_trace_chunk")

    def chunk_bounds(self, n, chunk, nchunks):
        lo, hi = ctypes.c_int64(), ctypes.c_int64()
        self._check(self.lib.rt_chunk_bounds(
            int(n), int(chunk), int(nchunks), ctypes.byref(lo),
            ctypes.byref(hi)), "rt_chunk_bounds")
        return lo.value, hi.value

    def download(self, which, lo, hi, out=None):
        """Rows [lo,hi) of one array as a compact SoA host array:
        (rows,3,N) for y/u/i, (rows,N) for t (into ``out`` if given)."""
        n = self.nrays
        shape = (hi - lo, n) if which == RT_T else (hi - lo, 3, n)
        if out is None:
            out = np.empty(shape, dtype=np.float64)
        elif out.shape != shape or not out.flags.c_contiguous \
                or out.dtype != np.float64:
            raise ValueError("download: `out` must be C-contiguous float64 "
                             "of shape %r" % (shape,))
        self._check(self.lib.rt_download(self.ctx, which, lo, hi,
                                         out.ctypes.data), "rt_download")
        return out

    def download_xy(self, which, surf, out=None):
        """x and y of one row of y / u / i: (2, N) (rt_download_xy)."""
        shape = (2, self.nrays)
        if out is None:
            out = np.empty(shape)
        elif out.shape != shape or not out.flags.c_contiguous \
                or out.dtype != np.float64:
            raise ValueError("download_xy: `out` must be C-contiguous "
                             "float64 of shape %r" % (shape,))
        self._check(self.lib.rt_download_xy(self.ctx, which, int(surf),
                                            out.ctypes.data),
                    "rt_download_xy")
        return out

    def download_ray(self, which, ray):
        """One ray across all surfaces: (L,3) for y/u/i, (L,) for t."""
        L = self.nsurf
        out = np.empty(L if which == RT_T else (L, 3))
        self._check(self.lib.rt_download_ray(self.ctx, which, int(ray),
                                             out.ctypes.data),
                    "rt_download_ray")
        return out

    def download_rays(self, which, ray0, stride, count):
        """Every ``stride``-th ray from ``ray0`` on, ``count`` of them,
        across all surfaces, gathered on the device: ``(L, count, 3)`` for
        y / u / i, ``(L, count)`` for t (rt_download_rays) -- a sample of a
        batch too large to bring down."""
        L, nc = self.nsurf, 1 if which == RT_T else 3
        out = np.empty((L, nc, int(count)))
        self._check(self.lib.rt_download_rays(
            self.ctx, which, int(ray0), int(stride), int(count),
            out.ctypes.data), "rt_download_rays")
        return out[:, 0] if which == RT_T else np.moveaxis(out, 1, 2)

    # -- device-side consumers ---------------------------------------------
    def set_weights(self, w):
        if w is None:
            self._check(self.lib.rt_set_weights(self.ctx, None),
                        "rt_set_weights")
            return
        w = np.ascontiguousarray(w, dtype=np.float64)
        if w.shape != (self.nrays,):
            raise ValueError("weights must have shape (nrays,)")
        self._check(self.lib.rt_set_weights(self.ctx, w.ctypes.data),
                    "rt_set_weights")

    def rms(self, surf, ref=-1):
        out = ctypes.c_double()
        self._check(self.lib.rt_rms(self.ctx, int(surf), int(ref),
                                    ctypes.byref(out)), "rt_rms")
        return out.value

    def row_rmax(self, surf):
        out = ctypes.c_double()
        self._check(self.lib.rt_row_rmax(self.ctx, int(surf),
                                         ctypes.byref(out)), "rt_row_rmax")
        return out.value

    def spot_stats(self, surf, group_rays, ngroups):
        """(ngroups, 6): count, centroid x y, weighted mean d^2, max d^2,
        sum w of every bundle of the batch (rt_spot_stats)."""
        out = np.empty((int(ngroups), 6))
        self._check(self.lib.rt_spot_stats(self.ctx, int(surf),
                                           int(group_rays), int(ngroups),
                                           out.ctypes.data), "rt_spot_stats")
        return out

    def row_stats(self, surf, group_rays, ngroups, ref=-1):
        """(ngroups, 10): count, sum w, mean x y, spread about the mean,
        spread about ray ``ref`` of the bundle, max x^2 + y^2, weighted
        centroid x y, spread about the shift -- one pass (rt_row_stats)."""
        out = np.empty((int(ngroups), 10))
        self._check(self.lib.rt_row_stats(
            self.ctx, int(surf), int(group_rays), int(ngroups),
            -1 if ref is None else int(ref), out.ctypes.data), "rt_row_stats")
        return out

    def refocus_shift(self, surf):
        out = ctypes.c_double()
        self._check(self.lib.rt_refocus_shift(self.ctx, int(surf),
                                              ctypes.byref(out)),
                    "rt_refocus_shift")
        return out.value

    def opd_rays(self, args):
        """(3, n): x, y on the reference sphere and t in waves."""
        args = np.ascontiguousarray(args, dtype=_lib.OPD_ARGS_DTYPE)
        # 24 B per ray into FRESH host pages cost more in page faults than in
        # PCIe (10^7 rays: 17 ms against 5): the buffer of the last call is
        # taken again when nobody holds it or a view of it any more
        # (references: this attribute, the local name, getrefcount's argument)
        out = getattr(self, "_opd_out", None)
        if out is None or out.shape != (3, self.nrays) or \
                sys.getrefcount(out) > 3:
            out = np.empty((3, self.nrays))
        self._opd_out = None        # (not kept if the call raises)
        self._check(self.lib.rt_opd_rays(self.ctx, args.ctypes.data,
                                         out.ctypes.data), "rt_opd_rays")
        self._opd_out = out
        return out

    def opd_stats(self, args, group_rays=None, ngroups=1, keep=False):
        """Per bundle ``[count, sum w, mean, rms about the mean, min, max,
        peak to valley, rms about the reference ray]`` of the optical path
        differences (waves), reduced on the device (rt_opd_stats): nothing
        per ray crosses PCIe.  ``args["ref"]`` is the reference ray's index
        inside a bundle; ``keep`` leaves x | y | t on the device
        (:meth:`opd_device`)."""
        args = np.ascontiguousarray(args, dtype=_lib.OPD_ARGS_DTYPE)
        if group_rays is None:
            group_rays = self.nrays//ngroups
        out = np.empty((int(ngroups), _lib.RT_OPD_STATS))
        self._check(self.lib.rt_opd_stats(
            self.ctx, args.ctypes.data, int(group_rays), int(ngroups),
            1 if keep else 0, out.ctypes.data), "rt_opd_stats")
        return out

    def opd_device(self):
        """(device address, rays) of the ``[3][rays]`` array x | y | t the
        last :meth:`opd_stats` (``keep=True``) or :meth:`opd_rays` left on
        the device; :meth:`copy_to_host` moves the part a caller plots."""
        p, n = ctypes.c_void_p(), ctypes.c_int64()
        self._check(self.lib.rt_opd_device(self.ctx, ctypes.byref(p),
                                           ctypes.byref(n)), "rt_opd_device")
        return p.value, n.value

    def device_ptr(self, which, surf):
        p = ctypes.c_void_p()
        self._check(self.lib.rt_device_ptr(self.ctx, which, surf,
                                           ctypes.byref(p)), "rt_device_ptr")
        return p.value

    def input_uniform(self):
        """(tiles uniform per launch component y0 y1 y2 u0 u1 u2, tiles): how
        much of row 0 a trace from element 1 reads (rt_input_uniform)."""
        t = (ctypes.c_int64*7)()
        self._check(self.lib.rt_input_uniform(self.ctx, t), "rt_input_uniform")
        return list(t[:6]), int(t[6])

    def input_completed(self):
        """Tiles of row 0 whose u_z a trace rebuilds from u_x, u_y instead of
        reading it (rt_input_completed)."""
        t = ctypes.c_int64()
        self._check(self.lib.rt_input_completed(self.ctx, ctypes.byref(t)),
                    "rt_input_completed")
        return int(t.value)

    def placement(self):
        """Where the result arrays live (rt_placement): dict with the pieces
        behind them (0 = plain hipMalloc), MiB per piece, pieces created on
        the way, classes seen, pieces kept per class, the batch's own store
        pattern over the arrays (GB/s; per set of pieces tried) and ``fast``:
        four workgroups per CU."""
        info = (ctypes.c_int*16)()
        ms = (ctypes.c_double*16)()
        self._check(self.lib.rt_placement(self.ctx, info, ms), "rt_placement")
        return {"pieces": info[0], "piece_mib": info[1], "created": info[2],
                "classes": info[3], "per_class": [info[4], info[5], info[6]],
                "fast": bool(info[7]), "ballast_blocks": info[8],
                "classes_mixed": bool(info[9]), "hops": info[8],
                "store_pattern_GBps": ms[2],
                "piece_sets_tried": info[12],
                "gave_up_incoherent": bool(info[13]),
                "store_pattern_GBps_per_piece_set": [
                    ms[8 + k] for k in range(max(min(info[12], 5), 0))],
                "pair_test_ms": {"same_piece": ms[0], "other_class": ms[1]},
                "search_ms": {"all": ms[3], "pieces": ms[4], "ballast": ms[5],
                              "remap": ms[6], "tune": ms[7],
                              "slowest_hipMemCreate": ms[13],
                              "reserve_total": ms[14]},
                "search_cut_short": {0: None, 1: "time budget",
                                     2: "hipMemCreate stalled"}.get(info[10]),
                "settled": bool(info[11]),
                "vm_call_failures_in_process": info[14],
                "orders_tried": info[15]}

    def selftest_arith(self, seed, n, span=100):
        """rt_selftest_arith: mismatch counts (refraction quotient, table
        quotient, square root, unguarded core)."""
        bad = (ctypes.c_uint64*4)()
        self._check(self.lib.rt_selftest_arith(self.ctx, int(seed), int(n),
                                               int(span), bad),
                    "rt_selftest_arith")
        return [int(b) for b in bad]

    def scratch(self, nbytes):
        p = ctypes.c_void_p()
        self._check(self.lib.rt_scratch(self.ctx, int(nbytes),
                                        ctypes.byref(p)), "rt_scratch")
        return p.value

    def copy_to_host(self, d_src, nbytes, dtype=np.float64):
        out = np.empty(nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self._check(self.lib.rt_copy_to_host(
            self.ctx, out.ctypes.data, ctypes.c_void_p(d_src), int(nbytes)),
            "rt_copy_to_host")
        return out

    # -- multi GPU ----------------------------------------------------------
    def comm_unique_id(self):
        buf = ctypes.create_string_buffer(128)
        rc = self.lib.rt_comm_unique_id(buf)
        if rc != 0:
            raise EngineError("rt_comm_unique_id failed (%d): %s" % (
                rc, (self.lib.rt_last_error(None) or b"").decode()))
        return buf.raw

    def comm_init(self, unique_id, nranks, rank):
        buf = ctypes.create_string_buffer(bytes(unique_id), 128)
        self._check(self.lib.rt_comm_init(self.ctx, buf, nranks, rank),
                    "rt_comm_init")

    def comm_info(self, max_devices=16):
        """What the communicator says about itself (rt_comm_info)."""
        info = (ctypes.c_int*4)()
        lt = (ctypes.c_int*max_devices)()
        hp = (ctypes.c_int*max_devices)()
        self._check(self.lib.rt_comm_info(self.ctx, info, lt, hp,
                                          max_devices), "rt_comm_info")
        nd = min(info[3], max_devices)
        names = {1: "HyperTransport", 2: "PCIe", 3: "InfiniBand", 4: "xGMI"}
        return {"nranks_seen": info[0], "rank_seen": info[1],
                "rccl_version_code": info[2], "devices_visible": info[3],
                "links": [{"device": d, "type": names.get(lt[d], lt[d]),
                           "hops": hp[d]} for d in range(nd)
                          if d != self.device]}

    def gather_final(self, which, surf, counts, root, d_dst):
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        self._check(self.lib.rt_gather_final(
            self.ctx, which, surf,
            counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), root,
            ctypes.c_void_p(d_dst)), "rt_gather_final")

    def gather_chunk(self, which, surf, counts, root, d_dst, chunk, nchunks):
        """:meth:`gather_final` for chunk ``chunk`` of ``nchunks`` of every
        rank's rays: overlaps the trace of the next chunk."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        self._check(self.lib.rt_gather_chunk(
            self.ctx, which, surf,
            counts.ctypes.data_as(ctypes.POINTER(ctypes.c_int64)), root,
            ctypes.c_void_p(d_dst), int(chunk), int(nchunks)),
            "rt_gather_chunk")

    def gather_ms(self):
        """(total, exposed) ms of the last gather from HIP events: exposed =
        what is left after this rank's last trace kernel finished."""
        total, exposed = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.rt_gather_ms(self.ctx, ctypes.byref(total),
                                          ctypes.byref(exposed)),
                    "rt_gather_ms")
        return total.value, exposed.value

    def comm_destroy(self):
        self._check(self.lib.rt_comm_destroy(self.ctx), "rt_comm_destroy")

    def comm_sync(self):
        self._check(self.lib.rt_comm_sync(self.ctx), "rt_comm_sync")


_engines = {}


def get_engine(device=None):
    """Process-wide engine per device (contexts are not thread safe)."""
    dev = _default_device() if device is None else int(device)
    eng = _engines.get(dev)
    if eng is None or eng.ctx is None:
        eng = _engines[dev] = Engine(dev)
    return eng


def march_rows(system, y, u, n, l, start=1, stop=None, clip=False,
               engine=None):
    """System.propagate (rayopt/system.py:459-464) as one fused GPU trace.

    ``y, u``: (N,3) in the global orientation relative to the vertex of
    element ``start-1`` (the caller has already applied that element's
    ``from_normal``, as geometric_trace.py:76 does), so the seed row is
    packed as unrotated.  Yields ``(y, u, n, i, t)`` per element.
    """
    eng = engine or get_engine()
    y, u = np.atleast_2d(y, u)
    y, u = np.broadcast_arrays(np.asarray(y, float), np.asarray(u, float))
    a, b = resolve_range(len(system), start, stop)
    if a < 1:
        raise ValueError("start must be >= 1")
    table, ns = pack_system(system, l, n, a, b)
    table["flags"][a - 1] &= ~np.uint32(_lib.F_ROTATED)
    eng.upload_system(table)
    eng.set_rays(y, u)
    if a > 1:   # seed row is start-1, not 0
        eng.upload_row(RT_Y, a - 1, np.ascontiguousarray(y.T))
        eng.upload_row(RT_U, a - 1, np.ascontiguousarray(u.T))
    eng.trace(a, b, clip)
    for j in range(a, b):
        yj = eng.download(RT_Y, j, j + 1)[0].T
        uj = eng.download(RT_U, j, j + 1)[0].T
        ij = eng.download(RT_I, j, j + 1)[0].T
        tj = eng.download(RT_T, j, j + 1)[0]
        yield yj, uj, ns[j], ij, tj


def element_propagate(element, y0, u0, n0, l, clip=True, engine=None):
    """``Element.propagate(y0, u0, n0, l, clip)`` (rayopt/elements.py:230-236,
    306-315) for one element on the GPU: ``y0, u0`` (N,3) already in the
    element's normal frame relative to its vertex (no transfer, no rotation:
    those belong to System.propagate).  Returns ``(y, u, n, t*n0)``."""
    from .model import Spheroid
    eng = engine or get_engine()
    y0, u0 = np.atleast_2d(y0, u0)
    y0, u0 = np.broadcast_arrays(np.asarray(y0, float), np.asarray(u0, float))
    intercept_only = l is None
    table, ns = pack_system([Spheroid(), element],
                            5.8756e-7 if intercept_only else l, n0)
    if intercept_only:      # geometry only: no index change, no bending
        table["flags"][1] &= ~np.uint32(_lib.F_REFRACT | _lib.F_MIRROR)
    table["offset"][1] = 0.
    table["rot"][1] = np.eye(3).reshape(9)
    table["flags"][1] &= ~np.uint32(_lib.F_ROTATED)
    eng.upload_system(table)
    eng.set_rays(np.ascontiguousarray(y0), np.ascontiguousarray(u0))
    eng.set_keep_rows(None)
    eng.trace(1, 2, clip)
    y = eng.download(RT_Y, 1, 2)[0].T
    u = eng.download(RT_U, 1, 2)[0].T
    t = eng.download(RT_T, 1, 2)[0]
    return y, u, ns[1], t
