"""``aiming="rayopt"``: the INSTALLED rayopt's own aiming methods bound to
device traces -- part of the drop-in glue (this package next to an importable
rayopt), like ``dropin.accelerate``; the self-contained form of the same
procedure is ``aiming="reference"`` (rayopt_amd/aiming_reference.py).

``FieldAimer`` (rayopt_amd/aiming.py) is this package's aimer: all fields in
one kernel, iterated to 1e-9.  The reference (rayopt/system.py:466-593) aims
one field at a time with scipy's secant and Brent solvers stopped at
``tol = 1e-3`` and seeds every field with an interpolation of the fields
solved before it (rayopt/cachend.py:84-105): its aimed pupils carry that
tolerance AND that history, and a bundle launched from them differs from one
launched from the exact pupil in the fourth digit.

Where a user needs rayopt's numbers -- regression data of an existing design,
the ``Analysis`` figures -- ``GeometricTrace(system, aiming="rayopt")``
runs **rayopt's own code**: this module contains no solver and no guess
cache.  It takes ``solve_newton``, ``solve_brentq``, ``aim_chief``,
``aim_marginal``, ``_aim_pupil`` and ``pupil`` from the *installed* rayopt
(the drop-in scenario: rayopt is importable next to this package), unchanged,
and binds them to a thin object whose ``aim()`` and ``propagate()`` -- the
only two things those methods call -- build the launch ray with the device's
generation kernel and trace it on the device.  With identical traced
intercepts (plane / sphere / conic systems: bit identical) the solvers take
identical iterates: aimed pupils equal rayopt's (tests/test_aiming_reference.py
asserts ``==`` for systems without tilted elements; a ONE-ray trace of a
tilted element goes through another BLAS routine in numpy -- gemv instead of
gemm -- than the kernel's FMA chain reproduces: 1e-10 there, INTEGRATION.md
section 1).  What this proves: the DEVICE TRACES under rayopt's solvers; it
says nothing about ``FieldAimer``.  Without an importable rayopt this option
raises ImportError; nothing is written into the rayopt module.
"""
import importlib
import sys
import types

import numpy as np

from ..aiming import pupil_option, start_pupil

_BORROWED = ("solve_newton", "solve_brentq", "aim_chief", "aim_marginal",
             "_aim_pupil", "pupil")


def _installed_rayopt():
    mod = sys.modules.get("rayopt")
    if mod is None:
        try:
            mod = importlib.import_module("rayopt")
        except ImportError as err:
            raise ImportError(
                "aiming='rayopt' binds the installed rayopt's own aiming "
                "methods (rayopt/system.py:466-593) to device traces and "
                "needs rayopt importable; aiming='reference' runs the same "
                "procedure without it, aiming='device' is this package's "
                "aimer (FieldAimer)") from err
    return mod


_BOUND = {}      # id(rayopt module) -> (module, class): kept HERE


def _bound_class(rayopt):
    """A class carrying the reference's aiming methods, unmodified, next to
    device-backed ``aim`` / ``propagate``; made once per rayopt module and
    kept in this module (nothing is written into rayopt)."""
    hit = _BOUND.get(id(rayopt))
    if hit is None or hit[0] is not rayopt:
        ref = rayopt.system.System
        body = {name: ref.__dict__[name] for name in _BORROWED}
        hit = _BOUND[id(rayopt)] = (
            rayopt, type("DeviceTracedSystem", (_DeviceTraced,), body))
    return hit[1]


class _DeviceTraced:
    """What rayopt's aiming methods see as ``self``: the user's System for
    everything they read, the device for the two things they compute."""

    def __init__(self, system, engine, l):
        self._system, self._engine = system, engine
        self._pupil_cache = {}
        self.evaluations = 0
        z0, r0 = start_pupil(system, l)
        # the object pupil as the last update() left it (rayopt/system.py:
        # 562-565 reads distance and radius from there)
        self.object = types.SimpleNamespace(
            finite=bool(system.object.finite), wideangle=False,
            pupil=types.SimpleNamespace(
                distance=z0, radius=r0,
                telecentric=bool(pupil_option(system, "telecentric")),
                aim=bool(pupil_option(system, "aim"))))

    wavelengths = property(lambda self: self._system.wavelengths)
    stop = property(lambda self: self._system.stop)

    def __len__(self):
        return len(self._system)

    def __getitem__(self, index):
        return self._system[index]

    def refractive_index(self, wavelength, index):
        return self._system.refractive_index(wavelength, index)

    def aim(self, *args, **kwargs):
        """``System.aim`` (rayopt/system.py:503-504): launch ray(s) from the
        device's generation kernel."""
        # (this package's System.aim, also for an unmodified rayopt System:
        # it reads public attributes only)
        from ..model import System
        return System.aim(self._system, *args, engine=self._engine, **kwargs)

    def propagate(self, y, u, n, l, start=1, stop=None, clip=False):
        """``System.propagate`` (rayopt/system.py:459-464) as one device
        trace, handed out element by element."""
        from ..engine import march_rows
        self.evaluations += 1
        return march_rows(self._system, y, u, n, l, start, stop, clip,
                          engine=self._engine)


class RayoptAimer:
    """``System.pupil(yo, l, stop)`` of the installed rayopt with the one-ray
    traces on this engine.  One instance per (system, wavelength, stop); keep
    it for as long as the prescription does not change -- like rayopt's
    ``_pupil_cache``, which ``System.update()`` clears."""

    def __init__(self, system, engine=None, l=None, stop=None, given=None):
        if engine is None:
            from ..engine import get_engine
            engine = get_engine()
        self.system, self.stop, self._given = system, stop, given
        self.l = system.wavelengths[0] if l is None else l
        self._traced = _bound_class(_installed_rayopt())(system, engine,
                                                         self.l)

    evaluations = property(lambda self: self._traced.evaluations)

    def chief(self, yo, z, p):
        """rayopt's ``aim_chief`` for field ``yo`` from pupil ``(z, p)``."""
        return self._traced.aim_chief(np.asarray(yo, dtype=float), z, p,
                                      l=self._given, stop=self.stop)

    def pupil(self, yo):
        """(z, a[2][2]) for field ``yo``; earlier answers seed later ones
        (rayopt's own guess cache)."""
        z, a = self._traced.pupil((float(yo[0]), float(yo[1])),
                                  l=self._given, stop=self.stop)
        return z, a.copy()
