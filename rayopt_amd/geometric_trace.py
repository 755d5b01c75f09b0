"""Drop-in ``GeometricTrace`` whose ``propagate()`` runs on the MI355X.

Interface and result arrays follow rayopt (rayopt/geometric_trace.py:29-80,
rayopt/raytrace.py:24-36):

    ``y[j]``  intercept at element j          (L,N,3)
    ``u[j]``  direction after element j       (L,N,3)
    ``i[j]``  direction before element j      (L,N,3)
    ``t[j]``  optical path of segment j       (L,N)
    ``n[j]``  refractive index after j        (L,)

all in the element-normal frame relative to the vertex; dead rays are NaN.
The arrays live in HBM (SoA ``[surface][component][ray]``) and are exposed
as lazy host views with the reference's shapes: a surface row is copied over
PCIe only when it is first touched (``trace.y[-1]``), and the view is a
strided transpose of the SoA buffer, not a re-layout.

``system`` may be a :class:`rayopt_amd.model.System` or an unmodified
reference ``rayopt.System``: only public element attributes are read
(rayopt_amd/pack.py).
"""
import numpy as np

from . import _lib
from ._lib import RT_Y, RT_U, RT_I, RT_T
from .engine import Engine, get_engine
from .pack import pack_system, pack_tables, resolve_range


class Trace:
    """rayopt/raytrace.py:24-36."""
    def __init__(self, system):
        self.system = system

    def allocate(self):
        self.length = len(self.system)

    def propagate(self):
        self.path = self.system.path
        self.track = self.system.track
        self.origins = self.system.origins
        self.mirrored = self.system.mirrored

    def from_axis(self, y, i=None, ref=0):
        """Points given along the folded optical axis -- ``y[k, ray]`` =
        (x, y, z) with z the path length along the axis -- in the global
        frame: every point is assigned to the element whose stretch of the
        axis it lies on (``i``: the split indices, by default found from ray
        ``ref``'s z against ``path``), shifted to that element's vertex and
        turned by its axis rotation (rayopt/raytrace.py:38-54)."""
        y = np.atleast_3d(y)
        if i is None:
            i = np.searchsorted(y[:, ref, 2], self.path)
        pieces = []
        for j, block in enumerate(np.vsplit(y, i)):
            if block.ndim <= 1:
                continue
            j = min(j, self.length - 1)
            flat = block.reshape(-1, 3) - (0., 0., self.path[j])
            flat = self.origins[j] + self.system[j].from_axis(flat)
            pieces.append(flat.reshape(block.shape))
        return np.vstack(pieces)

    def print_coeffs(self, coeff, labels, sum=True):
        """Lines of a per-element table: index, element type letter, one
        column per label; a totals row unless ``sum=False``
        (rayopt/raytrace.py:56-63)."""
        width = len(labels)
        yield ("%2s %1s" + "% 10s"*width) % (("#", "T") + tuple(labels))
        row = "%2s %1s" + "% 10.4g"*width
        for j, values in enumerate(coeff):
            letter = getattr(self.system[j], "typeletter", "S")
            yield row % ((j, letter) + tuple(values))
        if sum:
            yield row % (("", "") + tuple(np.sum(coeff, axis=0)))

    def align(self):
        """Tilt the elements for the axial ray given the indices of the last
        trace (``System.align``), then re-trace (rayopt/raytrace.py:65-67)."""
        self.system.align(self.n)
        self.propagate()


class DeviceRows:
    """Lazy host view of one device-resident result array.

    Behaves like the reference's ``np.ndarray`` of shape (L,N,3) / (L,N) for
    reading: indexing the first axis with an int or a unit-step slice copies
    just those surface rows from the GPU (once); anything else, and
    ``np.asarray``, materialises the whole array.
    """
    __array_priority__ = 100.

    def __init__(self, trace, which):
        self._trace = trace
        self._which = which
        self._host = None       # (L,3,N) or (L,N), allocated on first touch
        self._valid = np.zeros(trace.length, dtype=bool)
        self._valid_xy = np.zeros(trace.length, dtype=bool)
        self.dtype = np.dtype(np.float64)

    @property
    def shape(self):
        t = self._trace
        return ((t.length, t.nrays) if self._which == RT_T
                else (t.length, t.nrays, 3))

    ndim = property(lambda self: len(self.shape))
    size = property(lambda self: int(np.prod(self.shape)))

    def __len__(self):
        return self._trace.length

    def _buffer(self):
        if self._host is None:
            t = self._trace
            shape = ((t.length, t.nrays) if self._which == RT_T
                     else (t.length, 3, t.nrays))
            self._host = np.empty(shape)
        return self._host

    def invalidate(self, lo, hi):
        self._valid[lo:hi] = False
        self._valid_xy[lo:hi] = False

    @staticmethod
    def _only_xy(rest):
        """``rows[j, :, :2]`` and its spellings (``0``, ``1``, ``0:2`` on the
        last axis): the reference's spot consumers read nothing else."""
        if len(rest) != 2 or rest[0] != slice(None):
            return False
        c = rest[1]
        if isinstance(c, (int, np.integer)):
            return c in (0, 1)
        return isinstance(c, slice) and c.step in (None, 1) and \
            c.start in (None, 0, 1) and c.stop in (1, 2)

    def _view_xy(self, j):
        """Row j with x and y present (z too if the row is here already)."""
        host = self._buffer()
        if not (self._valid[j] or self._valid_xy[j]):
            self._trace._engine.download_xy(self._which, j, out=host[j, :2])
            self._valid_xy[j] = True
        return host[j].T

    def put_row(self, j, soa):
        """Host already knows this row (rays_given)."""
        self._buffer()[j] = soa
        self._valid[j] = True

    def _ensure(self, lo, hi):
        host = self._buffer()
        j = lo
        while j < hi:           # fetch maximal runs of missing rows
            if self._valid[j]:
                j += 1
                continue
            k = j
            while k < hi and not self._valid[k]:
                k += 1
            self._trace._engine.download(self._which, j, k, out=host[j:k])
            self._valid[j:k] = True
            j = k
        return host

    def _view(self, lo, hi):
        host = self._ensure(lo, hi)[lo:hi]
        return host if self._which == RT_T else host.transpose(0, 2, 1)

    def __getitem__(self, key):
        first, rest = (key[0], key[1:]) if isinstance(key, tuple) else (key, ())
        length = self._trace.length
        if isinstance(first, (int, np.integer)):
            j = int(first)
            if j < 0:
                j += length
            if not 0 <= j < length:
                raise IndexError("surface index %d out of range" % first)
            if self._which != RT_T and self._only_xy(rest):
                return self._view_xy(j)[rest]       # two thirds of the row
            out = self._view(j, j + 1)[0]
        elif isinstance(first, slice) and first.step in (None, 1):
            idx = range(length)[first]
            lo, hi = idx.start, max(idx.start, idx.stop)
            out = self._view(lo, hi) if hi > lo else \
                np.empty((0,) + self.shape[1:])
        else:
            out = self._view(0, length)[first]
        return out[rest] if rest else out

    def __array__(self, dtype=None, copy=None):
        out = self._view(0, self._trace.length)
        if dtype is not None and np.dtype(dtype) != out.dtype:
            return out.astype(dtype)
        return np.array(out) if copy else out

    def __setitem__(self, key, value):
        raise TypeError(
            "result arrays are device resident and read-only from the host; "
            "seed rays with rays_given()")

    def __iter__(self):
        for j in range(self._trace.length):
            yield self[j]

    def __repr__(self):
        return "<DeviceRows %s shape=%s valid=%d/%d>" % (
            "yuit"[self._which], self.shape, self._valid.sum(),
            len(self._valid))


class GeometricTrace(Trace):
    """
    y[i]: intercept at surface
    i[i]: incoming/incidence direction before surface
    u[i]: outgoing/excidence direction after surface
    all in i-surface normal coordinates relative to vertex
    """
    def __init__(self, system, engine=None, device=None, **options):
        """``options`` (extension): engine options applied to this trace's
        context (``rt_set_option``), e.g. ``exact_asphere=True`` -- even
        aspheres on the bit-for-bit restatement of scipy's Newton iteration
        (the reference's bits) instead of the default FMA / rcp / rsq
        arithmetic, whose results are within the 1e-8 contract for iterated
        aspheres -- or ``compact=1``, the clipped-ray compacting kernel for
        traces that do not store every row."""
        super().__init__(system)
        self._engine = engine
        self._device = device
        self._options = dict(options)
        # "device": all fields aimed by one kernel, iterated to 1e-9;
        # "reference": rayopt's procedure and tolerances, field by field, as
        # this package restates it (rayopt_amd/aiming_reference.py) -- for
        # numbers that must match rayopt's; "rayopt": the same with the
        # INSTALLED rayopt's own methods (rayopt_amd/dropin/aiming_rayopt.py)
        self._aiming = self._options.pop("aiming", "device")
        if self._aiming not in ("device", "reference", "rayopt"):
            raise ValueError(
                "aiming must be 'device', 'reference' or 'rayopt'")
        if engine is not None:
            self._apply_options()

    def _apply_options(self):
        for key, value in self._options.items():
            self._engine.set_option(key, int(value))

    # -- Trace.propagate (rayopt/raytrace.py:31-35), evaluated lazily ------
    # path / track / origins / mirrored are O(L) cumulative sums that cost as
    # much host time as packing the table; most propagate() calls (merit
    # evaluations, aiming iterations) never look at them.  propagate()
    # snapshots what they are computed from -- the elements stay mutable --
    # and the arrays are built on first access.
    _LAZY = ("path", "track", "origins", "mirrored")

    def _snapshot_geometry(self):
        system = self.system
        self._geometry = (
            [el.offset for el in system], [el.distance for el in system],
            [getattr(getattr(el, "material", None), "mirror", False)
             for el in system])
        d = self.__dict__
        for name in self._LAZY:
            d.pop(name, None)

    def __getattr__(self, name):
        # reached only when normal lookup fails
        if name in GeometricTrace._LAZY and "_geometry" in self.__dict__:
            offsets, distances, mirrors = self._geometry
            if name == "path":              # rayopt/system.py:423-424
                value = np.cumsum(distances)
            elif name == "mirrored":        # :439-442
                value = np.cumprod([-1 if m else 1 for m in mirrors])
            else:                           # :414-415, :427-428
                origins = self.__dict__.get("origins")
                if origins is None:
                    origins = np.cumsum(offsets, axis=0)
                    self.__dict__["origins"] = origins
                value = origins if name == "origins" else origins[:, 2]
            self.__dict__[name] = value
            return value
        raise AttributeError("%r object has no attribute %r" % (
            type(self).__name__, name))

    # -- storage ----------------------------------------------------------
    @property
    def engine(self):
        if self._engine is None:
            self._engine = Engine(self._device)   # raises without GPU/.so
            self._apply_options()
        return self._engine

    def allocate(self, nrays):
        super().allocate()
        self.nrays = nrays
        self.n = np.empty(self.length)
        self.w = None
        self.ref = None
        self.l = 1.
        self._reset_bundles()
        if nrays == 0:      # nothing to trace: plain empty host arrays
            self.y, self.u, self.i = (np.empty((self.length, 0, 3))
                                      for _ in range(3))
            self.t = np.empty((self.length, 0))
            return
        self.y = DeviceRows(self, RT_Y)
        self.u = DeviceRows(self, RT_U)
        self.i = DeviceRows(self, RT_I)
        self.t = DeviceRows(self, RT_T)

    def _aux_engine(self):
        """Engine for the small aiming batches: the process-wide context of
        this device rather than a new one per aimer (for a test double: a
        second instance of it)."""
        if isinstance(self.engine, Engine):
            return get_engine(self._device)
        return type(self.engine)()

    def _reset_bundles(self):
        """Forget the bundle layout of an earlier rays_points() and the
        system variants of an earlier rays_variants()."""
        self.rays_per_field = None
        self.rays_alive_per_field = None
        self.rays_per_group = None
        self._variants = None

    def _upload_table(self, start, stop, n_init, packed=None):
        """Pack + hand over the surface table(s): one per wavelength when
        ``self.l`` is a sequence (ray groups), returns (tables, n).
        ``packed``: tables a caller packed a moment ago for the same
        wavelength(s) and the full range (the aimer's), reused as they are."""
        variants = getattr(self, "_variants", None)
        if variants is not None:
            # the batch holds one group of rays per system variant
            packed = [pack_system(v, self.l, n0, start, stop)
                      for v, n0 in zip(variants, n_init)]
            table = np.stack([t for t, _ in packed])
            ns = np.stack([n for _, n in packed])
        elif packed is not None:
            table, ns = packed
            if np.ndim(self.l) == 0 and table.ndim == 2:
                table, ns = table[0], ns[0]
        elif np.ndim(self.l) == 0:
            table, ns = pack_system(self.system, self.l, n_init, start, stop)
        else:
            table, ns = pack_tables(self.system, self.l, n_init, start, stop)
        self.engine.upload_system(table)
        self._packed = (resolve_range(self.length, start, stop), ns)
        return table, ns

    # -- seeding ------------------------------------------------------------
    def rays_given(self, y, u, l=None, w=None, ref=0):
        """Seed surface 0 (rayopt/geometric_trace.py:49-70)."""
        y, u = np.atleast_2d(y, u)
        y, u = np.broadcast_arrays(y, u)
        if y.ndim != 2 or y.shape[1] not in (2, 3):
            raise ValueError("rays_given: y and u must broadcast to (N,2) or "
                             "(N,3), got %r" % (y.shape,))
        n, m = y.shape
        if n == 0:          # an empty batch is legal in the reference
            self.allocate(0)
            self.l = self.system.wavelengths[0] if l is None else l
            self.w = np.empty(0) if w is None else np.asarray(w)
            self.ref = ref
            self.n[0] = self.system.refractive_index(self.l, 0)
            return
        if not hasattr(self, "y") or self.nrays != n \
                or self.length != len(self.system):
            self.allocate(n)
        self._reset_bundles()
        if l is None:
            l = self.system.wavelengths[0]
        if np.ndim(l) == 1:
            return self._rays_given_groups(y, u, np.asarray(l, float), w, ref)
        if np.ndim(self.n) != 1:    # an earlier batch had one n per group
            self.n = np.empty(self.length)
        if w is not None and np.shape(w) != (n,):
            raise ValueError("rays_given: w must have shape (%d,)" % n)
        self._uniform_w = w is None
        if w is None:
            # same values as the reference's np.ones(n)/n, without the
            # 8 B/ray allocation (read-only broadcast view)
            w = np.broadcast_to(np.ones(1)/n, (n,))
        self.w = w
        self.ref = ref
        self.l = l
        if m == 3 and y.dtype == np.float64 and u.dtype == np.float64 \
                and y.flags.c_contiguous and u.flags.c_contiguous:
            y0, u0 = y, u                    # hand over as is, no host copy
        else:
            y0 = np.zeros((n, 3))
            y0[:, :m] = y
            u0 = np.empty((n, 3))
            u0[:, :m] = u
            if m < 3:  # assumes forward rays
                u2 = np.square(u0[:, :2]).sum(-1)
                u0[:, 2] = np.sqrt(1 - u2)
        self.n[0] = self.system.refractive_index(l, 0)
        self._upload_table(1, None, self.n[0])
        self.engine.set_rays(y0, u0)
        self.engine.set_weights(None if self._uniform_w else w)
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(0, self.length)   # row 0 is read back on demand

    def _rays_given_groups(self, y, u, wavelengths, w, ref):
        """``rays_given(y, u, l=[l1, l2, ...])`` (extension): the same P rays
        at W wavelengths in ONE trace.  The batch holds W groups of P rays,
        group g at ``wavelengths[g]``: ray ``g*P + p``; ``n`` becomes (W, L).
        P must be a multiple of 64 so each wavefront stays inside one group
        and keeps reading its surface table through scalar loads."""
        p, m = y.shape
        groups = len(wavelengths)
        if p % 64:
            raise ValueError("rays_given with several wavelengths needs a "
                             "multiple of 64 rays per wavelength, got %d" % p)
        n = p*groups
        if not hasattr(self, "y") or self.nrays != n \
                or self.length != len(self.system):
            self.allocate(n)
        y0 = np.zeros((p, 3))
        y0[:, :m] = y
        u0 = np.empty((p, 3))
        u0[:, :m] = u
        if m < 3:
            u0[:, 2] = np.sqrt(1 - np.square(u0[:, :2]).sum(-1))
        self.l = wavelengths
        self.rays_per_group = p
        self._uniform_w = w is None
        if w is None:
            w = np.broadcast_to(np.ones(1)/n, (n,))
        elif np.shape(w) == (p,):
            w = np.tile(np.asarray(w, float)/groups, groups)
        elif np.shape(w) != (n,):
            raise ValueError("rays_given: w must have shape (%d,) or (%d,)"
                             % (p, n))
        self.w = w
        self.ref = ref
        self.n = np.empty((groups, self.length))
        self.n[:, 0] = [self.system.refractive_index(l, 0)
                        for l in wavelengths]
        self._upload_table(1, None, self.n[:, 0])
        self.engine.set_rays_repeat(y0, u0, groups)
        self.engine.set_weights(None if self._uniform_w else w)
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(0, self.length)

    def rays_variants(self, y, u, variants, l=None, w=None, ref=0):
        """The same P rays through V variants of the system in ONE trace
        (extension): a tolerancing run (V perturbed copies), the points of a
        finite-difference gradient, a design family.  ``variants``: sequence
        of V systems with the same number of elements as ``self.system``
        (e.g. deep copies with perturbed curvatures, thicknesses, tilts,
        glasses); they are packed again on every ``propagate()``, like
        ``self.system`` is.  The batch holds V groups, ray ``v*P' + p`` is ray
        ``p`` in variant ``v``; ``P'`` = P padded with dead (NaN, weight 0)
        rays to a multiple of 64 (``rays_per_group``; ``rays_alive_per_group``
        = P).  ``n`` becomes (V, L); ``spot_stats()`` / ``rms_fields()`` give
        one row per variant."""
        y, u = np.atleast_2d(y, u)
        y, u = np.broadcast_arrays(y, u)
        p, m = y.shape
        variants = list(variants)
        nv = len(variants)
        if nv < 1 or any(len(v) != len(self.system) for v in variants):
            raise ValueError("rays_variants: every variant needs %d elements"
                             % len(self.system))
        pp = p + (-p % 64)
        n = pp*nv
        if not hasattr(self, "y") or self.nrays != n \
                or self.length != len(self.system):
            self.allocate(n)
        self._reset_bundles()
        y0 = np.full((pp, 3), np.nan)
        y0[:p] = 0.
        y0[:p, :m] = y
        u0 = np.full((pp, 3), np.nan)
        u0[:p, :m] = u
        if m < 3:
            u0[:p, 2] = np.sqrt(1 - np.square(u0[:p, :2]).sum(-1))
        self.l = self.system.wavelengths[0] if l is None else l
        wp = np.zeros(pp)
        wp[:p] = 1./p if w is None else w
        self.w = np.tile(wp, nv)
        self._uniform_w = False
        self.ref = ref
        self._variants = variants
        self.rays_per_group = pp
        self.rays_alive_per_field = p     # what rms_fields(lost=...) counts
        self.n = np.empty((nv, self.length))
        self.n[:, 0] = [v.refractive_index(self.l, 0) for v in variants]
        self._upload_table(1, None, self.n[:, 0])
        self.engine.set_rays_repeat(y0, u0, nv)
        self.engine.set_weights(self.w)
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(0, self.length)

    def rays_given_device(self, d_y, d_u, nrays, l=None, w=None, ref=0,
                          layout=_lib.LAYOUT_SOA):
        """Seed surface 0 from arrays already in device memory (raw device
        pointers, float64, (3,N) SoA or (N,3) AoS): no PCIe transfer."""
        if not hasattr(self, "y") or self.nrays != nrays \
                or self.length != len(self.system):
            self.allocate(nrays)
        self._reset_bundles()
        self.l = self.system.wavelengths[0] if l is None else l
        if np.ndim(self.l) != 0:
            raise ValueError("rays_given_device takes one wavelength")
        if np.ndim(self.n) != 1:
            self.n = np.empty(self.length)
        if w is not None and np.shape(w) != (nrays,):
            raise ValueError("rays_given_device: w must have shape (%d,)"
                             % nrays)
        self._uniform_w = w is None
        self.w = np.broadcast_to(np.ones(1)/nrays, (nrays,)) if w is None \
            else np.asarray(w, dtype=float)
        self.ref = ref
        self.n[0] = self.system.refractive_index(self.l, 0)
        self._upload_table(1, None, self.n[0])
        self.engine.set_rays_device(d_y, d_u, nrays, layout)
        # the device reductions (rms, refocus, spot_stats) read the weights
        # of THIS batch: upload them, or drop those of an earlier one
        self.engine.set_weights(None if self._uniform_w else self.w)
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(0, self.length)

    def rays_fields(self, yo, yp, z, a, l=None, ref=0, _packed=None):
        """Launch ``len(yo) x len(yp)`` rays built on the GPU: for every
        field point ``yo[f]`` (fractional object coordinates) the bundle
        through the pupil coordinates ``yp`` (P,2), as
        ``system.aim(yo[f], yp, z[f], a[f], filter=False)`` + ``rays_given``
        would (rayopt/system.py:504, rayopt/conjugates.py:137-166,236-255)
        -- ray ``f*P + p`` -- without the 48 B/ray host transfer.  ``z, a``:
        pupil distance and aperture per field (e.g. from ``system.pupil``).

        ``l`` may be a sequence of W wavelengths (extension): the batch then
        holds W groups of F bundles, ray ``(w*F + f)*P + p``, traced in one
        launch with one surface table per wavelength; ``z``, ``a`` are given
        per wavelength, (W,) / (W,F) and (W,F,2,2); ``n`` becomes (W, L).
        ``F*P`` must be a multiple of 64 (see :meth:`rays_points`, which
        pads the pupil pattern)."""
        from .launch import field_frames
        yp = np.atleast_2d(np.asarray(yp, dtype=float))
        if l is None:
            l = self.system.wavelengths[0]
        if np.ndim(l) == 0:
            fields = field_frames(self.system, yo, z, a)
            groups = 0
        else:
            l = np.asarray(l, dtype=float)
            groups = len(l)
            fields = field_frames(self.system, yo, z, a, groups)
            per = len(fields)//groups*yp.shape[0]
            if per % 64:
                raise ValueError(
                    "several wavelengths in one batch need a multiple of 64 "
                    "rays per wavelength, got %d" % per)
        nrays = len(fields)*yp.shape[0]
        if not hasattr(self, "y") or self.nrays != nrays \
                or self.length != len(self.system):
            self.allocate(nrays)
        self._reset_bundles()
        self.l = l
        self._uniform_w = True
        self.w = np.broadcast_to(np.ones(1)/nrays, (nrays,))
        self.ref = ref
        if groups:
            self.n = np.empty((groups, self.length))
            self.n[:, 0] = [self.system.refractive_index(li, 0) for li in l]
            self._upload_table(1, None, self.n[:, 0], _packed)
        else:
            self.n = np.empty(self.length)
            self.n[0] = self.system.refractive_index(l, 0)
            self._upload_table(1, None, self.n[0], _packed)
        self.engine.generate_rays(fields, yp)
        self.engine.set_weights(None)
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(0, self.length)

    def rays_points(self, fields, wavelength=None, nrays=11,
                    distribution="meridional", clip=False, aim=True,
                    rim=False, keep=None):
        """Bundles for many field points in one go -- the batched counterpart
        of ``rays_point`` (rayopt/geometric_trace.py:204-209): pupil pattern
        (``pupil_distribution``), aiming of every field on the GPU
        (:class:`rayopt_amd.aiming.FieldAimer`; ``aim=False`` uses the
        paraxial entrance pupil, ``aim=None`` does what the object pupil's
        own ``aim`` flag says, as the reference's ``rays_point`` would), ray
        construction on the GPU, trace.  Ray
        ``f*P + p`` belongs to field ``f``; ``self.w`` carries the quadrature
        weights of the pattern, normalised per field.

        ``wavelength`` may be a sequence of W wavelengths: every field is
        aimed and traced at every wavelength in the same launch, ray
        ``(w*F + f)*P + p``.  The pupil pattern is then padded with dead
        (NaN) rays of weight 0 to the next multiple of 64 so that every
        wavefront reads a single surface table; ``rays_per_field`` is the
        padded bundle size, ``rays_alive_per_field`` the pattern's.
        ``keep`` is handed to :meth:`propagate`."""
        from .aiming import FieldAimer, entrance_pupil
        from .pupil import pupil_distribution
        fields = np.atleast_2d(np.asarray(fields, dtype=float))
        ref, yp, weight = pupil_distribution(distribution, nrays)
        l = self.system.wavelengths[0] if wavelength is None else wavelength
        alive = len(yp)
        packed = None
        if np.ndim(l) == 0:
            if aim or aim is None:
                aimer = FieldAimer(self.system, l, self._aux_engine(),
                                   aim=aim)
                z, a = aimer.pupil(fields, rim=rim)
                packed = aimer.packed
            else:
                z, a = entrance_pupil(self.system, l)
            copies = len(fields)
        else:
            l = np.asarray(l, dtype=float)
            if aim or aim is None:
                aimer = FieldAimer(self.system, l[0], self._aux_engine(),
                                   aim=aim)
                z, a = aimer.pupils(fields, l, rim=rim)
                packed = aimer.packed
            else:
                za = [entrance_pupil(self.system, li) for li in l]
                z = [np.broadcast_to(zi, (len(fields),)) for zi, _ in za]
                a = [ai for _, ai in za]
            pad = -alive % 64
            if pad:
                yp = np.concatenate([yp, np.full((pad, 2), np.nan)])
                if weight is None:
                    weight = np.ones(alive)/alive
                weight = np.concatenate([weight, np.zeros(pad)])
            copies = len(fields)*len(l)
        self.rays_fields(fields, yp, z, a, l, ref=ref, _packed=packed)
        if weight is not None:
            self.w = np.tile(weight, copies)
            self._uniform_w = False
            self.engine.set_weights(self.w)
        self.fields = fields
        self.rays_per_field = len(yp)
        self.rays_alive_per_field = alive
        self.propagate(clip=clip, keep=keep, _fresh=True)

    def spot_stats(self, i=-1, group_rays=None):
        """Spot statistics of every bundle of the batch at surface ``i`` in
        one device reduction (``rt_spot_stats``): array (..., 6) with
        ``count, centroid x, centroid y, sum(w d^2)/sum(w), max d^2, sum(w)``
        over the rays of the bundle that arrived; leading shape (F,) after
        :meth:`rays_points`, (W, F) with several wavelengths."""
        if group_rays is None:
            group_rays = self.rays_per_field or self.rays_per_group or \
                self.nrays
        groups, rest = divmod(self.nrays, int(group_rays))
        if rest:
            raise ValueError("bundles of %d rays do not tile %d rays"
                             % (group_rays, self.nrays))
        out = self.engine.spot_stats(range(self.length)[i], group_rays, groups)
        if np.ndim(self.l) == 1 and groups % len(self.l) == 0:
            out = out.reshape(len(self.l), -1, 6)
        return out

    ROW_STATS = np.dtype([(k, "f8") for k in (
        "count", "sum_w", "mean_x", "mean_y", "var_mean", "var_ref", "r2_max",
        "centroid_x", "centroid_y", "var_shift")])

    def row_stats(self, i=-1, group_rays=None, ref=None):
        """Everything :meth:`rms` (about the mean and about the reference
        ray), :meth:`spot_stats` and :meth:`resize` ask of row ``i``, for
        every bundle of the batch, in ONE pass over the row
        (``rt_row_stats``; rayopt/geometric_trace.py:171-193): structured
        array, one record per bundle (leading shape as :meth:`spot_stats`)
        with ``count, sum_w, mean_x, mean_y, var_mean`` (= rms()**2 of the
        bundle), ``var_ref`` (= rms(ref=...)**2; ``ref``: index inside a
        bundle, default ``self.ref``), ``r2_max`` (resize: radius**2),
        ``centroid_x, centroid_y`` (weighted)."""
        if group_rays is None:
            group_rays = self.rays_per_field or self.rays_per_group or \
                self.nrays
        groups, rest = divmod(self.nrays, int(group_rays))
        if rest:
            raise ValueError("bundles of %d rays do not tile %d rays"
                             % (group_rays, self.nrays))
        if ref is None:
            ref = self.ref if self.ref is not None else -1
        out = self.engine.row_stats(range(self.length)[i], group_rays, groups,
                                    ref).view(self.ROW_STATS)[:, 0]
        if np.ndim(self.l) == 1 and groups % len(self.l) == 0 and \
                len(self.l) > 1:
            out = out.reshape(len(self.l), -1)
        return out

    def rms_fields(self, i=-1, lost="nan"):
        """RMS spot radius of every bundle about its own centroid: what
        ``rays_point(field) ; rms()`` gives per field (and wavelength) in the
        reference (rayopt/geometric_trace.py:171-183,204-209), for the whole
        batch at once.  ``lost``: "nan" = a bundle that lost a ray gives NaN
        as the reference does; "omit" = statistics of the rays that
        arrived."""
        s = self.row_stats(i, ref=-1)       # one pass over the row
        r = np.sqrt(s["var_mean"])
        if lost == "nan":
            alive = self.rays_alive_per_field or \
                self.nrays//max(1, s["count"].size)
            r = np.where(s["count"] < alive, np.nan, r)
        elif lost != "omit":
            raise ValueError("lost must be 'nan' or 'omit'")
        return r

    def _pupil(self, yo, l, rim=False, given=None):
        """(z (1,), a (1,2,2)) of one field, by the configured aiming;
        ``given``: the caller's wavelength argument, None if defaulted."""
        if self._aiming != "device":
            from .aiming_reference import reference_aimer
            z, a = reference_aimer(self.system, self._aux_engine(), l,
                                   -1 if rim else None, given,
                                   self._aiming).pupil(yo)
            return np.array([z]), a[None]
        from .aiming import FieldAimer
        return FieldAimer(self.system, l, self._aux_engine(),
                          aim=None).pupil([yo], rim=rim)

    def rays(self, yo, yp, wavelength=None, stop=None, filter=None,
             clip=False, weight=None, ref=0):
        """One field point ``yo`` through the pupil coordinates ``yp`` (P,2):
        aiming (``stop=-1``: to the rim of the limiting aperture), optional
        filtering of ``yp`` to the aimed pupil ellipse (``Pupil.map(
        filter=True)``, rayopt/pupils.py:97-107; default ``not clip``), ray
        construction on the GPU, trace (rayopt/geometric_trace.py:195-202)."""
        from .aiming import FieldAimer
        if filter is None:
            filter = not clip
        yp = np.atleast_2d(np.asarray(yp, dtype=float))
        l = self.system.wavelengths[0] if wavelength is None else wavelength
        z, a = self._pupil(yo, l, rim=(stop == -1), given=wavelength)
        if filter:
            # Pupil.map(filter=True) (rayopt/pupils.py:97-107) acts on the
            # aperture as the conjugate hands it over: angles atan2(a, z) for
            # an object at finite distance (rayopt/conjugates.py:146)
            af = np.arctan2(a[0], z[0]) if self.system.object.finite \
                else a[0]
            am = np.fabs(af).max()
            c = np.sum(af, axis=0)/2
            d = np.diff(af, axis=0)/2
            inside = (np.square(yp*am - c)/np.square(d)).sum(1) <= 1
            yp = yp[inside]
            if weight is not None:
                weight = np.asarray(weight)[inside]
            ref = int(np.count_nonzero(inside[:ref]))
        self.rays_fields([yo], yp, z, a, l, ref=ref)
        if weight is not None:
            self.w = weight
            self._uniform_w = False
            self.engine.set_weights(weight)
        self.propagate(clip=clip, _fresh=True)

    def rays_point(self, yo, wavelength=None, nrays=11,
                   distribution="meridional", filter=None, stop=None,
                   clip=False):
        """One field point with a pupil sampling pattern, as the reference's
        ``rays_point`` (rayopt/geometric_trace.py:204-209)."""
        from .pupil import pupil_distribution
        ref, yp, weight = pupil_distribution(distribution, nrays)
        self.rays(yo, yp, wavelength, filter=filter, stop=stop, clip=clip,
                  weight=weight, ref=ref)

    def rays_clipping(self, yo, wavelength=None, axis=1):
        """Chief ray and the two rays grazing the limiting apertures along
        ``axis`` (rayopt/geometric_trace.py:211-215)."""
        from .aiming import FieldAimer
        l = self.system.wavelengths[0] if wavelength is None else wavelength
        z, a = self._pupil(yo, l, rim=True, given=wavelength)
        yp = np.zeros((3, 2))
        yp[1:, axis] = a[0][:, axis]/np.fabs(a[0]).max()
        self.rays_fields([yo], yp, z, a, l)
        self.propagate(_fresh=True)

    def rays_line(self, yo, wavelength=None, nrays=21, eps=1e-2):
        """``nrays`` field points from the axis to ``yo``, three rays each:
        chief, and the neighbours at pupil coordinate ``eps`` along the
        meridional and the sagittal direction; ray ``k*nrays + f`` is ray
        kind ``k`` (0 chief, 1 meridional, 2 sagittal) of field ``f``
        (rayopt/geometric_trace.py:217-229).  The chief rays of all fields
        are aimed together on the GPU."""
        from .aiming import FieldAimer
        l = self.system.wavelengths[0] if wavelength is None else wavelength
        fields = np.linspace(0, 1, nrays)[:, None]*np.atleast_2d(yo)
        e = np.zeros((3, 2))
        e[(1, 2), (1, 0)] = eps
        if self._aiming != "device":
            # field after field, each chief ray started from the previous
            # field's pupil distance (rayopt/geometric_trace.py:223-226)
            from .aiming_reference import reference_aimer
            aimer = reference_aimer(self.system, self._aux_engine(), l, None,
                                    wavelength, self._aiming)
            zk, p = aimer.pupil((0., 0.))
            z, a = [], p[None]
            for field in fields:
                zk = aimer.chief(field, zk, np.fabs(p).max())
                z.append(zk)
            z = np.array(z)
        else:
            aimer = FieldAimer(self.system, l, self._aux_engine(), aim=None)
            z0, a = aimer.pupil([(0., 0.)])
            z = aimer.chief(fields, z0[0], np.fabs(a[0]).max())
        # built field-major on the device, re-ordered kind-major (a few
        # dozen rays: the reference's layout is part of the contract)
        self.rays_fields(fields, e, z, a[0], l)
        y0 = np.asarray(self.y[0]).reshape(nrays, 3, 3)
        u0 = np.asarray(self.u[0]).reshape(nrays, 3, 3)
        self.rays_given(y0.transpose(1, 0, 2).reshape(-1, 3),
                        u0.transpose(1, 0, 2).reshape(-1, 3), l)
        self.propagate()

    def rays_paraxial(self, wavelength=None, axis=1):
        """The paraxial marginal and chief ray traced as real rays
        (rayopt/geometric_trace.py:185-193 with ParaxialTrace.rays,
        rayopt/paraxial_trace.py:66-79): launched from the object with the
        first-order pupil (``entrance_pupil``; a specified object pupil
        radius is kept) -- ray 0 marginal, ray 1 chief."""
        from .aiming import start_pupil
        l = self.system.wavelengths[0] if wavelength is None else wavelength
        z, r = start_pupil(self.system, l)
        n0 = self.system.refractive_index(l, 0)
        obj = self.system.object
        if obj.finite:
            heights = (0., -obj.radius)
            slopes = (n0*r/z, n0*obj.radius/z)
        else:
            c = np.tan(obj.angle)
            heights = (r, -c*z)
            slopes = (0., n0*c)
        y = np.zeros((2, 2))
        y[:, axis] = heights
        u = np.zeros((2, 2))
        tan = np.array(slopes)
        u[:, axis] = tan/np.sqrt(1 + np.square(tan))      # sinarctan
        self.rays_given(y, u, l)
        self.propagate()

    def plot(self, ax, axis=1, **kwargs):
        """Ray paths in the global frame, ``axis`` against z
        (rayopt/geometric_trace.py:236-240)."""
        style = dict(color="green")
        style.update(kwargs)
        rows = np.asarray(self.y)                       # (L, N, 3), host
        path = np.empty_like(rows)
        for j, element in enumerate(self.system):
            path[j] = element.from_normal(rows[j]) + self.origins[j]
        ax.plot(path[..., 2], path[..., axis], **style)

    # -- the hot path ---------------------------------------------------------
    def propagate(self, start=1, stop=None, clip=False, keep=None,
                  _fresh=False, chunks=1, after_chunk=None):
        """Trace elements ``start .. stop-1`` for all rays on the GPU
        (rayopt/geometric_trace.py:72-80 + rayopt/system.py:459-464).

        ``keep`` (extension, default None = every row as in the reference):
        iterable of surface indices whose rows are stored, e.g. ``keep=[-1]``
        for the image-plane intercepts only.  The other rows are traced but
        not written (no HBM traffic); reading them raises.

        ``chunks, after_chunk`` (extension, multi-GPU jobs): the rays are
        traced in ``chunks`` pieces (``rt_trace_chunk``) and
        ``after_chunk(k, chunks)`` is called after piece ``k`` has been
        queued -- the place to start the gather of that piece
        (``Engine.gather_chunk``), which then runs on the communication
        stream while the next piece is traced.  The result is the result of
        the unchunked call."""
        if not hasattr(self, "y"):
            raise ValueError("propagate: no rays; call rays_given() first")
        self._snapshot_geometry()
        if len(self.system) != self.length:
            raise ValueError("the system changed length since rays_given()")
        a, b = resolve_range(self.length, start, stop)
        if a < 1:
            raise ValueError("start must be >= 1")
        if self.nrays == 0:     # only the indices are there to be filled in
            self.n[a:b] = pack_system(self.system, self.l, self.n[a - 1],
                                      a, b)[1][a:b]
            return
        grouped = np.ndim(self.l) == 1 or self._variants is not None
        packed, self._packed = getattr(self, "_packed", None), None
        if _fresh and packed is not None and packed[0] == (a, b):
            # called by a compound method right after its own seeding: the
            # table that was packed for it is the one on the device
            ns = packed[1]
        else:
            _, ns = self._upload_table(
                a, b, self.n[:, a - 1] if grouped else self.n[a - 1])
            self._packed = None
        if keep is None:
            self.engine.set_keep_rows(None)
        else:
            mask = np.zeros(self.length, dtype=np.uint8)
            mask[[range(self.length)[k] for k in keep]] = 1
            mask[:a] = 1        # rows before `start` are not touched
            self.engine.set_keep_rows(mask)
        if chunks <= 1:
            self.engine.trace(a, b, clip)
            if after_chunk is not None:
                after_chunk(0, 1)
        else:
            for k in range(chunks):
                self.engine.trace_chunk(a, b, clip, k, chunks)
                if after_chunk is not None:
                    after_chunk(k, chunks)
        if grouped:
            self.n[:, a:b] = ns[:, a:b]
        else:
            self.n[a:b] = ns[a:b]
        for rows in (self.y, self.u, self.i, self.t):
            rows.invalidate(a, b)

    def kernel_ms(self):
        """HIP-event duration of the last propagate() kernel."""
        return self.engine.kernel_ms()

    # -- consumers: O(N) reductions on the device ---------------------------
    def rms(self, i=-1, ref=None):
        """RMS spot radius at surface ``i`` about the centroid (or ray
        ``ref``), weighted with ``w`` (rayopt/geometric_trace.py:171-183);
        reduced on the GPU, two scalars cross PCIe."""
        if self.nrays == 0:
            return 0.       # sqrt of an empty sum, as in the reference
        return self.engine.rms(range(self.length)[i],
                               -1 if ref is None else range(self.nrays)[ref])

    def refocus(self, at=-1):
        """Least-squares refocus: moves element ``at`` along the axis so the
        weighted spot is smallest and re-traces
        (rayopt/geometric_trace.py:82-99).  The sums are reduced on the GPU."""
        t = self.engine.refocus_shift(range(self.length)[at])
        self.system[at].distance += t
        self.propagate()
        return t

    def resize(self, fn=lambda a, b: a):
        """Set every element's aperture from the largest ray height on it:
        ``radius = fn(max hypot(x, y), radius)``
        (rayopt/geometric_trace.py:231-234); the maxima are reduced on the
        GPU, one scalar per surface crosses PCIe."""
        for j, el in enumerate(self.system[1:], 1):
            el.radius = fn(self.engine.row_rmax(j), el.radius)

    def print_trace(self, rays=None):
        """Text table per ray: refractive index, track, path relative to the
        axial path, intercept, direction per element
        (rayopt/geometric_trace.py:242-255).  ``rays`` (extension): the ray
        indices to print, default all as in the reference; a ray costs three
        strided reads of L values from the device."""
        if np.ndim(self.l):
            raise NotImplementedError("one wavelength per table")
        labels = ("n/track z/rel path/height x/height y/height z/"
                  "angle x/angle y/angle z").split("/")
        for k in (range(self.nrays) if rays is None else rays):
            y = self.engine.download_ray(RT_Y, k)
            u = self.engine.download_ray(RT_U, k)
            t = np.cumsum(self.engine.download_ray(RT_T, k)) - self.path
            yield "ray %i" % k
            yield from self.print_coeffs(
                np.column_stack([self.n, self.path, t, y, u]), labels,
                sum=False)
            yield ""

    def text(self):
        """rayopt/geometric_trace.py:257-258."""
        return self.print_trace()

    def __str__(self):
        return "\n".join(self.text())

    def _image_pupil(self):
        """(telecentric, distance) of the image-side pupil: the reference's
        Pupil object as its paraxial update left it, or -- this package's
        conjugates -- the distance the last ``System.update()`` stored there
        (or the user pinned), else the paraxial exit pupil at the system's
        first wavelength evaluated now (rayopt/paraxial_trace.py:326-341)."""
        pupil = self.system.image.pupil
        if not isinstance(pupil, dict):
            return bool(pupil.telecentric), pupil.distance
        telecentric = bool(pupil.get("telecentric", False))
        if pupil.get("distance") is not None or telecentric:
            return telecentric, pupil.get("distance")
        from .aiming import exit_pupil
        return False, exit_pupil(self.system)[0]

    def opd_rays(self, radius=None, after=-2, image=-1):
        """Per-ray optical path difference on the reference sphere centred
        on the image of ray ``ref``: ``x, y`` (exit-pupil coordinates) and
        ``t`` in waves -- what the reference's ``opd(resample=0)`` returns
        (rayopt/geometric_trace.py:101-131), computed by one fused kernel."""
        x, y, t = self.engine.opd_rays(self._opd_args(radius, after, image))
        return x, y, t

    OPD_STATS = ("count", "sum_w", "mean", "rms", "min", "max", "pv",
                 "rms_about_ref")

    def opd_stats(self, radius=None, after=-2, image=-1, bundles=1,
                  keep=False):
        """Weighted mean / rms / peak-to-valley of the optical path
        differences (waves) of every bundle, reduced on the device: the
        per-ray values of :meth:`opd_rays` (rayopt/geometric_trace.py:
        101-131, before the reference resamples) never cross PCIe.  The
        batch is ``bundles`` contiguous bundles of equal size, each measured
        against its own reference ray (``ref`` counts inside a bundle).
        Returns an array ``(bundles, 8)``, columns :attr:`OPD_STATS`;
        ``keep=True`` leaves x | y | t on the device
        (``engine.opd_device()``)."""
        per = self.nrays//bundles
        if per*bundles != self.nrays:
            raise ValueError("opd_stats: %d rays do not split into %d "
                             "bundles" % (self.nrays, bundles))
        args = self._opd_args(radius, after, image, per)
        return self.engine.opd_stats(args, per, bundles, keep)

    def _opd_args(self, radius, after, image, bundle_rays=None):
        if np.ndim(self.l):
            raise NotImplementedError("opd of a multi-wavelength batch: "
                                      "trace one wavelength per trace")
        L = self.length
        nrows = len(range(L)[:after + 1])
        after, image = range(L)[after], range(L)[image]
        if radius is None:
            telecentric, distance = self._image_pupil()
            if telecentric:
                radius = self.track[image] - self.track[after]
            else:
                if distance is None:
                    raise ValueError("opd: give `radius` or an image pupil "
                                     "distance")
                radius = -distance
        ea, ei = self.system[after], self.system[image]
        args = np.zeros((), dtype=_lib.OPD_ARGS_DTYPE)
        args["nrows"], args["after"], args["image"] = nrows, after, image
        args["finite"] = bool(self.system.object.finite)
        args["ref"] = range(bundle_rays or self.nrays)[self.ref]
        args["n0"], args["n_after"] = self.n[0], self.n[after]
        args["radius"] = radius
        args["lscale"] = self.l/self.system.scale
        args["shift"] = self.origins[after] - self.origins[image]
        eye = np.eye(3)
        args["rot_after"] = bool(ea.rotated)
        args["rot_image"] = bool(ei.rotated)
        args["r_after"] = (ea.rot_normal if ea.rotated else eye).reshape(9)
        args["r_image"] = (ei.rot_normal if ei.rotated else eye).reshape(9)
        return args

    def opd(self, radius=None, after=-2, image=-1, resample=4):
        """OPD map: the per-ray values (device) optionally resampled onto a
        regular ``resample*sqrt(N)`` grid (host, scipy griddata) --
        rayopt/geometric_trace.py:101-144."""
        from .wavefront import resample_pupil
        x, y, t = self.opd_rays(radius, after, image)
        if resample:
            return resample_pupil(x, y, t, int(resample*self.nrays**.5))
        return x, y, t

    def psf(self, pad=4, resample=4, **kwargs):
        """Point spread function from the resampled pupil OPD (host FFT of
        the n x n grid) -- rayopt/geometric_trace.py:146-169."""
        from .wavefront import psf_from_opd
        if not resample:
            raise NotImplementedError
        radius = self.system[-1].distance
        x, y, o = self.opd(resample=resample, radius=radius, **kwargs)
        return psf_from_opd(x, o, pad, radius, self.l/self.system.scale)
