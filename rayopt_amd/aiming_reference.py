"""Aiming the way the reference does it -- for results that have to MATCH
rayopt's, not merely agree with them.

``FieldAimer`` (rayopt_amd/aiming.py) is this package's aimer: all fields in
one kernel, iterated to 1e-9.  The reference (rayopt/system.py:466-593) aims
one field at a time with scipy's secant and Brent solvers stopped at
``tol = 1e-3`` and seeds every field with an interpolation of the fields
solved before it (rayopt/cachend.py:84-105): its aimed pupils carry that
tolerance AND that history, and a bundle launched from them differs from one
launched from the exact pupil in the fourth digit.

Where a user needs rayopt's numbers -- regression data of an existing design,
the ``Analysis`` figures -- ``GeometricTrace(system, aiming="reference")``
runs **rayopt's own procedure**: this module contains no solver and no guess
cache.  It takes ``solve_newton``, ``solve_brentq``, ``aim_chief``,
``aim_marginal``, ``_aim_pupil`` and ``pupil`` from the *installed* rayopt
(the drop-in scenario: rayopt is importable next to this package), unchanged,
and binds them to a thin object whose ``aim()`` and ``propagate()`` -- the
only two things those methods call -- build the launch ray with the device's
generation kernel and trace it on the device.  With identical traced
intercepts (plane / sphere / conic systems: bit identical) the solvers take
identical iterates: aimed pupils equal rayopt's (tests/test_aiming_reference.py
asserts ``==``).  Without an importable rayopt this option raises; the native
answer is ``aiming="device"``.
"""
import importlib
import sys
import types

import numpy as np

from .aiming import pupil_option, start_pupil

_BORROWED = ("solve_newton", "solve_brentq", "aim_chief", "aim_marginal",
             "_aim_pupil", "pupil")


def _installed_rayopt():
    mod = sys.modules.get("rayopt")
    if mod is None:
        try:
            mod = importlib.import_module("rayopt")
        except ImportError as err:
            raise ImportError(
                "aiming='reference' runs rayopt's own aiming procedure "
                "(rayopt/system.py:466-593) on device traces and needs rayopt "
                "importable; this package's aimer is aiming='device' "
                "(FieldAimer)") from err
    return mod


def _bound_class(rayopt):
    """A class carrying the reference's aiming methods, unmodified, next to
    device-backed ``aim`` / ``propagate``; made once per rayopt module."""
    cls = rayopt.__dict__.get("_mi355_device_traced_system")
    if cls is None:
        ref = rayopt.system.System
        body = {name: ref.__dict__[name] for name in _BORROWED}
        cls = type("DeviceTracedSystem", (_DeviceTraced,), body)
        rayopt._mi355_device_traced_system = cls
    return cls


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
        from .model import System
        return System.aim(self._system, *args, engine=self._engine, **kwargs)

    def propagate(self, y, u, n, l, start=1, stop=None, clip=False):
        """``System.propagate`` (rayopt/system.py:459-464) as one device
        trace, handed out element by element."""
        from .engine import march_rows
        self.evaluations += 1
        return march_rows(self._system, y, u, n, l, start, stop, clip,
                          engine=self._engine)


class ReferenceAimer:
    """``System.pupil(yo, l, stop)`` of the installed rayopt with the one-ray
    traces on this engine.  One instance per (system, wavelength, stop); keep
    it for as long as the prescription does not change -- like rayopt's
    ``_pupil_cache``, which ``System.update()`` clears."""

    def __init__(self, system, engine=None, l=None, stop=None, given=None):
        if engine is None:
            from .engine import get_engine
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


def reference_aimer(system, engine, l, stop, given=None):
    """The aimer of (system, wavelength, stop), kept on the system like the
    reference's ``_pupil_cache`` and dropped by ``System.update()``.
    ``given`` is the wavelength argument as the caller passed it (``None``
    for "the default"): the reference keys its cache on that
    (rayopt/system.py:586), so a bundle requested without a wavelength and
    one requested at the first wavelength have separate guess histories."""
    home = getattr(system, "_pupil_cache", None)
    if isinstance(home, dict):
        # a rayopt System: live inside its own cache, which its update()
        # clears (rayopt/system.py:201-202), under a key of our own
        cache = home.setdefault("rayopt_amd reference aimers", {})
    else:
        cache = system.__dict__.setdefault("_reference_aimers", {})
    key = (given, stop)
    if key not in cache:
        cache[key] = ReferenceAimer(system, engine, l, stop, given)
    return cache[key]
