"""Batched ray aiming on the GPU.

The reference aims one field point at a time: ``System.pupil`` ->
``_aim_pupil`` -> ``aim_chief`` (secant on the chief ray's stop intercept) and
four ``aim_marginal`` solves (Brent on the marginal rays' aperture margin),
each evaluation a trace of ONE ray (rayopt/system.py:507-593; ~130 serial
N=1 traces per field).  Here all F field points are aimed by ONE kernel: a
lane owns a field and runs the five root finds -- launch frame, one-ray
trace, solver update -- in registers (``rt_aim_pupil``,
rayopt_amd/csrc/rt_aim.h); a few hundred microseconds whatever F is.  The
same solvers also exist as host loops in which every iteration is one device
trace of all fields (``on_device=False``; ~150 launches): the cross-check of
the kernel, and the form an engine without ``aim_pupil`` gets.  The roots are
the same
(the chief ray through the centre of the stop; marginal rays grazing the stop
edge, or the first limiting aperture for ``rim=True``); the reference stops
at ``tol=1e-3``, this solver iterates to ``tol`` (default 1e-9), so the two
agree to the reference's tolerance.

``entrance_pupil`` gives the first-order starting values the reference takes
from its paraxial trace (``ParaxialTrace.update_conjugates``,
rayopt/paraxial_trace.py:326-334): the paraxial image of the stop in object
space.
"""
import numpy as np

from .geometric_trace import GeometricTrace
from .launch import _telecentric


def _element_matrix(el, n0, l):
    """(index behind the element, its 4x4 first-order matrix acting on
    (x, y, n u_x, n u_y)): a transfer over the element's distance, then --
    for a surface -- refraction or reflection at incidence ``angles[0]`` in
    the form of Massey and Siegman (Appl. Opt. 8, 975) and the turn by
    ``angles[2]`` about the axis, entry by entry and product by product as
    the reference forms them (rayopt/elements.py:223-228, 300-304, 503-542):
    the first-order pupils below are the reference's to the last bit, and
    with them every aimed bundle of ``aiming="reference"``."""
    transfer = np.eye(4)
    transfer[0, 2] = transfer[1, 3] = el.distance/n0
    material = getattr(el, "material", None)
    n = n0 if material is None else el.refractive_index(l)
    if not hasattr(el, "curvature"):
        return n, transfer
    c = el.curvature
    if el.aspherics is not None:
        c = c + 2*el.aspherics[0]
    angles = np.asarray(el.angles, dtype=float)
    cos_i = np.cos(angles[0])
    bend = np.eye(4)
    if material is not None:
        if material.mirror:
            bend[2, 0] = 2*c*cos_i
            bend[3, 1] = 2*c/cos_i
        else:
            mu = n/n0
            p = np.sqrt(mu**2 + cos_i**2 - 1)
            bend[1, 1] = p/(mu*cos_i)
            bend[2, 0] = n0*c*(cos_i - p)
            bend[3, 1] = mu*bend[2, 0]/(cos_i*p)
            bend[3, 3] = 1/bend[1, 1]
    m = np.dot(bend, transfer)
    cos_t, sin_t = np.cos(angles[2]), np.sin(angles[2])
    turn = np.eye(4)
    turn[:2, :2] = turn[2:, 2:] = np.array([[cos_t, -sin_t],
                                            [sin_t, -cos_t]])
    return n, np.dot(turn, np.dot(m, turn.T))


def first_order_matrix(system, l, start=1, stop=None):
    """(index behind the last element, product of the element matrices of
    ``system[start:stop]``) -- rayopt/system.py:399-410."""
    n = system.refractive_index(l, start - 1)
    m = np.eye(4)
    for el in system[start:stop]:
        n, mi = _element_matrix(el, n, l)
        m = np.dot(mi, m)
    return n, m


def entrance_pupil(system, l=None):
    """(distance from the vertex of element 0, radius) of the first-order
    image of the stop in object space, from the meridional 2x2 block of the
    matrix object -> stop: what ``ParaxialTrace.update_conjugates`` hands
    ``system.object`` (rayopt/paraxial_trace.py:326-334)."""
    if l is None:
        l = system.wavelengths[0]
    stop = system.stop
    _, m = first_order_matrix(system, l, stop=stop + 1)
    a, b = m[1::2, 1::2][0]
    b *= system.refractive_index(l, 0)
    return b/a, system[stop].radius/a


def exit_pupil(system, l=None):
    """(distance from the vertex of the image surface, radius) of the
    first-order image of the stop in image space: first row of the inverse
    of the meridional block stop -> image -- what the reference stores in
    ``system.image.pupil`` (rayopt/paraxial_trace.py:336-341) and
    ``GeometricTrace.opd`` takes its default reference-sphere radius from
    (rayopt/geometric_trace.py:113)."""
    if l is None:
        l = system.wavelengths[0]
    stop = system.stop
    n, m = first_order_matrix(system, l, start=stop + 1)
    a, b = np.linalg.inv(m[1::2, 1::2])[0]
    b *= n
    return b/a, system[stop].radius/a


def start_pupil(system, l, z0=None, a0=None):
    """Starting pupil (distance, aperture): what the reference reads from
    ``system.object.pupil`` (rayopt/system.py:562-565) -- the distance the
    last ``update()`` stored (or the user pinned), else the paraxial image
    of the stop evaluated now; a specified object pupil radius is the
    starting aperture, as in the reference (``Pupil.update`` only tracks it
    if ``update_radius``), else the paraxial one.  ``l`` is the wavelength
    of the on-the-fly evaluation."""
    if z0 is None or a0 is None:
        spec = getattr(system.object, "pupil", None)
        get = spec.get if isinstance(spec, dict) else \
            (lambda key: getattr(spec, key, None))
        zp, given = get("distance"), get("radius")
        if isinstance(spec, dict) and zp is not None:
            from .design import pupil_radius    # slope / na / fno pupils
            given = pupil_radius(spec)
        if (z0 is None and zp is None) or (a0 is None and not given):
            zq, ap = entrance_pupil(system, l)
            zp = zq if zp is None else zp
            given = given or ap
        z0 = zp if z0 is None else z0
        a0 = given if a0 is None else a0
    return z0, a0


def pupil_option(system, name, default=False):
    """``system.object.pupil.<name>`` for this package's conjugates (a dict)
    and the reference's (a Pupil object, rayopt/pupils.py:29-38)."""
    pupil = getattr(system.object, "pupil", None)
    if isinstance(pupil, dict):
        return pupil.get(name, default)
    return getattr(pupil, name, default)


class FieldAimer:
    """Aims F field points at once.  ``engine`` is injectable (tests).

    ``aim``: None = as the reference decides (rayopt/system.py:507-531): the
    chief ray is aimed only if ``object.pupil.aim`` is set and the pupil is
    not telecentric, the marginal rays if ``aim`` is set or the rim is asked
    for; an object pupil without the flag -- the reference's default --
    launches from the first-order pupil as it is.  True = aim whatever the
    flag says (the batched extension entry points)."""

    def __init__(self, system, l=None, engine=None, tol=1e-9, maxiter=60,
                 on_device=True, aim=True):
        self.system = system
        self.aim = aim
        self.l = system.wavelengths[0] if l is None else l
        self.trace = GeometricTrace(system, engine=engine)
        self.tol = tol
        self.maxiter = maxiter
        self.on_device = on_device
        self.packed = None      # (tables, n) of the last device aiming

    # one trace of F rays: field f through pupil point yp with (z_f, a_f)
    def _stop_xy(self, yo, yp, z, a, last):
        t = self.trace
        t.rays_fields(yo, [yp], z, a, self.l)
        t.propagate(stop=last + 1)
        return t

    def chief(self, yo, z0, p, stop=None):
        """Pupil distance per field such that the chief ray crosses the
        centre of the stop (aim_chief, rayopt/system.py:507-526)."""
        yo = np.atleast_2d(np.asarray(yo, dtype=float))
        stop = self.system.stop if stop in (-1, None) else stop
        rad = self.system[stop].radius
        z0 = np.broadcast_to(np.asarray(z0, dtype=float), (len(yo),))
        todo = ~np.all(np.isclose(yo, 0), axis=1)   # on-axis: nothing to aim
        if not self._aims()[0]:                     # system.py:509-510
            todo[:] = False
        if not todo.any():
            return z0.copy()

        def miss(alpha):
            t = self._stop_xy(yo, (0., 0.), z0 + alpha*p, p, stop)
            y = np.asarray(t.y[stop])[:, :2]
            return (yo*y).sum(1)/rad

        a0 = np.zeros(len(yo))
        f0 = miss(a0)
        a1 = a0 + 1e-4
        f1 = miss(a1)
        for _ in range(self.maxiter):
            with np.errstate(all="ignore"):
                step = np.where(todo & (f1 != f0),
                                f1*(a1 - a0)/(f1 - f0), 0.)
            a0, f0 = a1, f1
            a1 = a1 - step
            if np.all(np.abs(step) <= self.tol):
                break
            f1 = miss(a1)
        else:
            raise ValueError("chief-ray aiming did not converge")
        return np.where(todo, z0 + a1*p, z0)

    def marginal(self, yo, yp, z, p, rim=False, stop=None):
        """Signed aperture per field such that the ray through pupil
        coordinate ``yp`` grazes the stop edge -- or, ``rim=True``, the first
        limiting aperture of the whole system (aim_marginal,
        rayopt/system.py:528-555).  Bracketing + regula falsi (Illinois),
        vectorised over fields."""
        yo = np.atleast_2d(np.asarray(yo, dtype=float))
        nf = len(yo)
        last = len(self.system) - 2 if rim else \
            (self.system.stop if stop is None else stop)
        r2 = np.square([e.radius for e in self.system[1:last + 1]])
        p = np.broadcast_to(np.asarray(p, dtype=float), (nf,))

        def margin(scale):
            t = self._stop_xy(yo, yp, z, np.abs(scale*p), last)
            if rim:
                ys = np.asarray(t.y[1:last + 1])[:, :, :2]
                return (np.square(ys).sum(2)/r2[:, None] - 1).max(0)
            y = np.asarray(t.y[last])[:, :2]
            return np.square(y).sum(1)/r2[-1] - 1

        lo = np.zeros(nf)
        flo = margin(np.full(nf, 1e-9))       # ~ chief ray: inside, < 0
        hi = np.ones(nf)
        for _ in range(self.maxiter):         # expand until outside
            fhi = margin(hi)
            bad = np.isnan(fhi)
            inside = ~bad & (fhi < 0)
            if not (bad | inside).any():
                break
            lo = np.where(inside, hi, lo)
            flo = np.where(inside, fhi, flo)
            hi = np.where(bad, hi/2, np.where(inside, hi*(1 - fhi), hi))
        else:
            raise ValueError("no viable marginal-ray interval")
        side = np.zeros(nf)
        for _ in range(self.maxiter):
            with np.errstate(all="ignore"):
                x = (lo*fhi - hi*flo)/(fhi - flo)
            x = np.where(np.isfinite(x), x, (lo + hi)/2)
            fx = margin(x)
            neg = fx < 0
            # Illinois: halve the retained end's value when it is kept twice
            flo = np.where(neg, fx, np.where(side == 1, flo/2, flo))
            fhi = np.where(neg, np.where(side == -1, fhi/2, fhi), fx)
            lo = np.where(neg, x, lo)
            hi = np.where(neg, hi, x)
            side = np.where(neg, -1., 1.)
            if np.all(np.abs(fx) <= self.tol):
                break
        else:
            raise ValueError("marginal-ray aiming did not converge")
        return x*p

    _FAILURES = {1: "chief-ray aiming did not converge",
                 2: "no viable marginal-ray interval",
                 3: "marginal-ray aiming did not converge"}

    def _start(self, l, z0=None, a0=None):
        return start_pupil(self.system, l, z0, a0)

    def _aims(self, rim=False):
        """(aim the chief ray, aim the marginal rays)."""
        flag = bool(pupil_option(self.system, "aim")) if self.aim is None \
            else bool(self.aim)
        telecentric = bool(pupil_option(self.system, "telecentric"))
        return flag and not telecentric, flag or bool(rim)

    @staticmethod
    def _unaimed(yo, z0, a0):
        """The first-order pupil for every field (_aim_pupil with both
        solvers returning their input, system.py:557-583)."""
        z = np.broadcast_to(np.asarray(z0, dtype=float), (len(yo),)).copy()
        r = np.broadcast_to(np.fabs(np.asarray(a0, dtype=float)), (len(yo),))
        a = r[:, None, None]*np.array([[-1., -1.], [1., 1.]])
        return z, a

    def _pupil_on_device(self, yo, wavelengths, starts, rim):
        """All fields at all wavelengths, all five root finds, one kernel
        (rt_aim_pupil): z (W,F), a (W,F,2,2)."""
        from ._lib import AIM_ARGS_DTYPE
        from .launch import aim_seeds
        from .pack import pack_system
        system = self.system
        packs = [pack_system(system, l, system.refractive_index(l, 0))
                 for l in wavelengths]
        tables = np.stack([t for t, _ in packs])
        self.packed = (tables, np.stack([n for _, n in packs]))
        engine = self.trace.engine
        engine.upload_system(tables)
        args = np.zeros((), dtype=AIM_ARGS_DTYPE)
        args["stop"], args["rim"] = system.stop, bool(rim)
        args["maxiter"], args["tol"] = self.maxiter, self.tol
        args["no_chief"] = not self._aims(rim)[0]
        seeds = aim_seeds(system, yo, [z0 for z0, _ in starts],
                          [a0 for _, a0 in starts], range(len(starts)))
        z, a, status = engine.aim_pupil(seeds, args)
        if status.any():
            bad = int(np.flatnonzero(status)[0])
            raise ValueError("%s (field %d: %r)" % (
                self._FAILURES.get(int(status[bad]), "aiming failed"),
                bad % len(yo), tuple(yo[bad % len(yo)])))
        groups = len(wavelengths)
        return z.reshape(groups, -1), a.reshape(groups, -1, 2, 2)

    def pupils(self, yo, wavelengths, rim=False):
        """:meth:`pupil` for every field at every wavelength -- z (W,F),
        a (W,F,2,2) -- in one launch where the engine aims on the device."""
        yo = np.atleast_2d(np.asarray(yo, dtype=float))
        starts = [self._start(l) for l in wavelengths]
        if not self._aims(rim)[1]:
            out = [self._unaimed(yo, z0, a0) for z0, a0 in starts]
            return (np.array([z for z, _ in out]),
                    np.array([a for _, a in out]))
        if self.on_device and hasattr(self.trace.engine, "aim_pupil"):
            return self._pupil_on_device(yo, wavelengths, starts, rim)
        keep = self.l
        try:
            out = []
            for l, (z0, a0) in zip(wavelengths, starts):
                self.l = l
                out.append(self.pupil(yo, z0, a0, rim))
        finally:
            self.l = keep
        return np.array([z for z, _ in out]), np.array([a for _, a in out])

    def pupil(self, yo, z0=None, a0=None, rim=False):
        """(z (F,), a (F,2,2)) for every field: chief aiming, then the four
        marginal rays -sag, -mer, +sag, +mer (_aim_pupil,
        rayopt/system.py:557-583; a = [[-sag,-mer],[+sag,+mer]])."""
        yo = np.atleast_2d(np.asarray(yo, dtype=float))
        nf = len(yo)
        z0, a0 = self._start(self.l, z0, a0)
        if not self._aims(rim)[1]:
            return self._unaimed(yo, z0, a0)
        if self.on_device and np.ndim(z0) == 0 and np.ndim(a0) == 0 \
                and hasattr(self.trace.engine, "aim_pupil"):
            z, a = self._pupil_on_device(yo, [self.l],
                                         [(float(z0), float(a0))], rim)
            return z[0], a[0]
        z = self.chief(yo, z0, np.fabs(a0))
        a = np.empty((nf, 2, 2))
        for axis in (1, 0):
            for sign in (1, 0):
                yp = [0., 0.]
                yp[axis] = 2*sign - 1.
                a[:, sign, axis] = (2*sign - 1.)*np.fabs(
                    self.marginal(yo, yp, z, a0, rim=rim))
        return z, a
