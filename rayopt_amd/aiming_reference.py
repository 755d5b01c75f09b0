"""Aiming the way the reference does it -- for results that have to MATCH
rayopt's, not merely agree with them.

``FieldAimer`` (rayopt_amd/aiming.py) aims all fields in one kernel and
iterates to 1e-9; the reference (rayopt/system.py:466-593) aims one field at a
time with scipy's secant and Brent solvers stopped at ``tol = 1e-3`` and
seeds every field with an interpolation of the fields solved before it
(``PolarCacheND``, rayopt/cachend.py:84-105).  Its aimed pupils therefore
carry that tolerance AND that history, and a bundle launched from them differs
from one launched from the exact pupil in the fourth digit.  Where a user
needs rayopt's numbers -- regression data of an existing design, the
``Analysis`` figures -- this module runs the reference's procedure itself:
the same starting values, the same two scipy solvers with the same
tolerances, the same guess cache and call order; only the function under the
solvers changes -- the one-ray trace to the stop (or through every aperture)
runs on the device (``rays_fields`` + ``propagate``: ~130 tiny launches per
field).  Select it with ``GeometricTrace(system, aiming="reference")`` or
``ReferenceAimer(system, engine).pupil(yo)``.  Self-contained: scipy, this
package, no rayopt.  (``aiming="rayopt"`` -- rayopt_amd/dropin/aiming_rayopt.py
-- binds the INSTALLED rayopt's own methods to device traces instead: the
cross-check of this restatement, tests/test_aiming_reference.py.)

With identical traced intercepts (plane / sphere / conic systems: bit
identical) the solvers take identical iterates; what is
left is the launch frame, which this package builds to 1e-13 of the
reference's (tests/test_generate.py), so aimed pupils agree to ~1e-12
(tests/test_aiming_reference.py).
"""
import numpy as np
from scipy.optimize import brentq, newton

from .aiming import pupil_option, start_pupil
from .geometric_trace import GeometricTrace


class PolarGuesses:
    """Solutions by field radius; a new field is seeded with the linear
    interpolation between its neighbours in |field|, clamped at the ends
    (rayopt/cachend.py:27-52, 84-105)."""
    def __init__(self):
        self.solved = {}

    def __contains__(self, key):
        return key in self.solved

    def guess(self, xo, yo):
        if not self.solved:
            return None
        keys = list(self.solved)
        radius = np.sqrt(np.square(np.array(keys)).sum(1))
        order = np.argsort(radius)
        radius = radius.take(order)
        values = np.array([self.solved[k] for k in keys]).take(order, axis=0)
        r = np.sqrt(xo**2 + yo**2)
        if r <= radius[0]:
            found = values[0]
        elif r >= radius[-1]:
            found = values[-1]
        else:
            i = np.searchsorted(radius, r)
            found = values[i - 1] + (values[i] - values[i - 1])*(
                r - radius[i - 1])/(radius[i] - radius[i - 1])
        return None if np.any(np.isnan(found)) else found


class ReferenceAimer:
    """``System.pupil(yo, l, stop)`` of the reference, on this engine.  One
    instance per (system, wavelength, stop); keep it for as long as the
    prescription does not change -- like the reference's ``_pupil_cache``,
    which ``System.update()`` clears."""

    def __init__(self, system, engine=None, l=None, stop=None, tol=1e-3,
                 maxiter=30):
        self.system = system
        self.l = system.wavelengths[0] if l is None else l
        self.stop = stop
        self.tol, self.maxiter = tol, maxiter
        self.trace = GeometricTrace(system, engine=engine)
        self.guesses = PolarGuesses()
        self.evaluations = 0

    # -- the function under the solvers: one ray on the device ---------------
    def _heights(self, yo, yp, z, a, upto):
        """x, y of the ray (field yo, pupil point yp, pupil z, a) on elements
        1 .. upto-1 (what ``self.propagate(y, u, n, l, stop=upto)`` yields)."""
        t = self.trace
        t.rays_fields([yo], [yp], z, a, self.l)
        t.propagate(stop=upto)
        self.evaluations += 1
        return np.asarray(t.y[1:upto])[:, 0, :2]

    # -- the reference's two solver front ends (system.py:466-500) -----------
    def _secant(self, merit, start=0.):
        value = merit(start)
        if np.isnan(value):             # look for a ray that gets through
            for scale in np.arange(1, self.maxiter):
                hit = [(start + s, merit(start + s)) for s in (-scale, scale)]
                hit = [(x, f) for x, f in hit if not np.isnan(f)]
                if hit:
                    start, value = hit[0]
                    break
            else:
                raise ValueError("no starting ray found")
        if abs(value) > self.tol:
            start = newton(merit, start, tol=self.tol, maxiter=self.maxiter)
        return start

    def _bracket(self, merit, lo=0., hi=1.):
        for trial in range(self.maxiter):
            f_hi = merit(hi)
            if abs(f_hi) <= self.tol:
                return hi
            if np.isnan(f_hi):
                hi /= 2
            elif f_hi < 0:
                lo, hi = hi, hi*(1 - f_hi)
            else:
                break
        if trial == self.maxiter - 1:
            raise ValueError("no viable interval found", lo, hi, f_hi)
        f_lo = merit(lo)
        if abs(f_lo) <= self.tol:
            return lo
        assert f_lo < 0
        return brentq(merit, lo, hi, rtol=self.tol, xtol=self.tol,
                      maxiter=self.maxiter)

    # -- aim_chief / aim_marginal (system.py:507-555) -------------------------
    def chief(self, yo, z, p):
        if pupil_option(self.system, "telecentric") or \
                not pupil_option(self.system, "aim"):
            return z
        stop = self.system.stop
        rad = self.system[stop].radius
        seen = {}

        def miss(a):
            if a not in seen:
                y = self._heights(yo, (0., 0.), z + a*p, abs(p), stop + 1)
                seen[a] = (yo*y[-1]).sum()/rad
            return seen[a]
        return z + self._secant(miss)*p

    def marginal(self, yo, yp, z, p):
        rim = self.stop == -1
        if not pupil_option(self.system, "aim") and not rim:
            return p
        upto = len(self.system) - 1 if rim else self.system.stop + 1
        r2 = np.square([e.radius for e in self.system[1:upto]])
        seen = {}

        def margin(a):
            if a not in seen:
                ys = self._heights(yo, yp, z, abs(a*p), upto)
                d = np.square(ys).sum(1)/r2 - 1
                seen[a] = d.max() if rim else d[-1]
            return seen[a]
        return self._bracket(margin)*p

    # -- _aim_pupil / pupil (system.py:557-593) -------------------------------
    def _solve(self, xo, yo, guess):
        y = np.array((xo, yo))
        if guess is None:
            z, r = start_pupil(self.system, self.l)
            a = r*np.ones((2, 2))
        else:
            z, a = guess[0], np.array(guess[1:], dtype=float).reshape(2, 2)
        if not np.allclose(y, 0):
            z1 = self.chief(y, z, np.fabs(a).max())
            if self.system.object.finite:
                a *= np.fabs(z1/z)
            z = z1
        for axis, sign in (1, 1), (1, 0), (0, 1), (0, 0):
            yp = [0, 0]
            yp[axis] = 2*sign - 1.
            a[sign, axis] = self.marginal(y, yp, z, a[sign, axis])
            if sign == 1:
                a[0, axis] = -a[1, axis]
            if (sign, axis) == (1, 1) and guess is None:
                a[:, 0] = a[:, 1]
        return np.r_[z, a.flat]

    def pupil(self, yo):
        """(z, a[2][2]) for field ``yo``; earlier answers seed later ones."""
        key = (float(yo[0]), float(yo[1]))
        if key not in self.guesses:
            self.guesses.solved[key] = self._solve(
                key[0], key[1], self.guesses.guess(*key))
        q = self.guesses.solved[key]
        return q[0], q[1:].reshape(2, 2).copy()


def reference_aimer(system, engine, l, stop, given=None, kind="reference"):
    """The aimer of (system, wavelength, stop), kept on the system like the
    reference's ``_pupil_cache`` and dropped by ``System.update()``.
    ``given`` is the wavelength argument as the caller passed it (``None``
    for "the default"): the reference keys its cache on that
    (rayopt/system.py:586), so a bundle requested without a wavelength and
    one requested at the first wavelength have separate guess histories.
    ``kind``: "reference" = this module's restatement, "rayopt" = the
    installed rayopt's own methods (rayopt_amd/dropin/aiming_rayopt.py;
    ImportError without rayopt)."""
    home = getattr(system, "_pupil_cache", None)
    if isinstance(home, dict):
        # a rayopt System: live inside its own cache, which its update()
        # clears (rayopt/system.py:201-202), under a key of our own
        cache = home.setdefault("rayopt_amd reference aimers", {})
    else:
        cache = system.__dict__.setdefault("_reference_aimers", {})
    key = (given, stop, kind)
    if key not in cache:
        if kind == "rayopt":
            from .dropin.aiming_rayopt import RayoptAimer
            cache[key] = RayoptAimer(system, engine, l, stop, given)
        else:
            cache[key] = ReferenceAimer(system, engine, l, stop)
    return cache[key]
