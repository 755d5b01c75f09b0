"""TEST INFRASTRUCTURE ONLY -- in-place import of the *unmodified* reference.

This module imports quartiq/rayopt from ``/root/reference`` without copying or
modifying it, applying the compatibility shims the reference needs on this
image:

* ``fastcache`` is not installed -> provide ``clru_cache`` on top of
  ``functools.lru_cache`` (used at rayopt/system.py:23, rayopt/material.py:22).
* ``rayopt.simplex_accel`` (Cython, PolyTrace only) is not built -> stub.
* PyYAML 6 requires an explicit ``Loader`` (rayopt/formats.py:86).
* ``np.complex_`` was removed in numpy 2 (rayopt/gaussian_trace.py:39).

It is used by ``tests/golden/make_golden.py`` (to generate the committed
golden fixtures), by the tests that compare the oracle and the device with
the real reference, and by ``bench.py``'s ``cpu_baseline`` leg.  The source is
``/root/reference`` where that exists (the build container) and otherwise the
byte-for-byte copy ``oracle/make_ref.py`` made under the git-ignored
``oracle/_ref/``, which travels to the GPU box like a built ``.so``.  It is
never imported by the product package (``rayopt_amd``).
"""
import functools
import os
import sys
import types

# where the unmodified reference lies: the tree itself in the build container,
# the byte-for-byte copy oracle/make_ref.py put under oracle/_ref/ (git-ignored,
# travels with the gpurun snapshot) on a GPU box
_CARRIED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref",
                        "rayopt_reference.zip")
REFERENCE_ROOT = os.environ.get("RAYOPT_REFERENCE", "/root/reference")
if not os.path.isdir(os.path.join(REFERENCE_ROOT, "rayopt")) and \
        os.path.isfile(_CARRIED):
    REFERENCE_ROOT = _CARRIED           # a zip on sys.path is importable


def carried():
    """True when the reference in use is the archive under oracle/_ref/."""
    return REFERENCE_ROOT == _CARRIED


def available():
    return carried() or os.path.isdir(os.path.join(REFERENCE_ROOT, "rayopt"))


def library_db():
    """Path of the reference's glass database (read-only; copy before use)."""
    if carried():
        return os.path.join(os.path.dirname(_CARRIED), "library.sqlite")
    return os.path.join(REFERENCE_ROOT, "rayopt", "library.sqlite")


def load():
    """Return the reference ``rayopt`` package (imported in place)."""
    if "rayopt" in sys.modules and getattr(
            sys.modules["rayopt"], "_amd_refshim", False):
        return sys.modules["rayopt"]
    if not available():
        raise ImportError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True      # the reference tree is read-only

    fc = types.ModuleType("fastcache")

    def clru_cache(maxsize=128, typed=False, **kw):
        return functools.lru_cache(maxsize=maxsize, typed=typed)
    fc.clru_cache = clru_cache
    sys.modules.setdefault("fastcache", fc)

    sa = types.ModuleType("rayopt.simplex_accel")
    sa.__all__ = ["simplex_transform", "simplex_mul", "simplex_pow",
                  "simplex_eval", "finite_object_fast"]
    for name in sa.__all__:
        setattr(sa, name, None)
    sys.modules["rayopt.simplex_accel"] = sa

    import yaml
    if not getattr(yaml.load, "_amd_refshim", False):
        _load = yaml.load

        def load_default(stream, Loader=yaml.Loader):
            return _load(stream, Loader=Loader)
        load_default._amd_refshim = True
        yaml.load = load_default

    import numpy as np
    if not hasattr(np, "complex_"):
        np.complex_ = np.complex128

    import warnings
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        import rayopt
    rayopt._amd_refshim = True
    return rayopt
