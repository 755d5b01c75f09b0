"""Swap the MI355X engine into an installed rayopt.

``accelerate(rayopt)`` returns (and installs) a ``GeometricTrace`` class whose
storage and ``propagate()`` are this package's (device resident, HIP kernel)
while the ray-generation helpers that only *call* ``rays_given``/``propagate``
-- ``rays, rays_point, rays_line, rays_clipping, rays_paraxial, resize, plot,
print_trace`` (rayopt/geometric_trace.py:185-259) -- are taken from the
installed rayopt unchanged, so aiming, pupils and conjugates keep running
rayopt's own host code.  ``rayopt.GeometricTrace`` and the name imported by
``rayopt.analysis`` are rebound, so ``Analysis`` traces on the GPU too.
"""
import numpy as np

from .geometric_trace import GeometricTrace

BORROWED = ("rays", "rays_point", "rays_line", "rays_clipping",
            "rays_paraxial", "resize", "plot", "print_trace")


class LegacyArray(np.ndarray):
    """ndarray with the methods numpy 2 dropped that rayopt's Analysis still
    calls on trace results (``ptp``, rayopt/analysis.py:314)."""
    def ptp(self, *args, **kwargs):
        return np.ptp(np.asarray(self), *args, **kwargs)


def modernize(rayopt=None):
    """Let an unmodified rayopt run on current numpy / matplotlib: restore
    the aliases and the no-op axis method it still uses (``np.int``,
    ``np.complex_``, ``np.float_``; ``Axis.set_smart_bounds``, removed in
    matplotlib 3.4; rayopt/special_sums.py:149, gaussian_trace.py:39,
    analysis.py:160-161).  Idempotent; touches nothing that exists."""
    for name, value in (("int", int), ("float_", np.float64),
                        ("complex_", np.complex128)):
        try:
            getattr(np, name)
        except AttributeError:
            setattr(np, name, value)
    try:
        from matplotlib.axis import Axis
    except ImportError:
        return
    if not hasattr(Axis, "set_smart_bounds"):
        Axis.set_smart_bounds = lambda self, value: None


def accelerate(rayopt, install=True, engine_factory=None):
    ref_cls = rayopt.geometric_trace.GeometricTrace
    if getattr(ref_cls, "_mi355", False):
        return ref_cls
    namespace = {"_mi355": True, "_reference_class": ref_cls,
                 "__doc__": GeometricTrace.__doc__}
    for name in BORROWED:
        if hasattr(ref_cls, name):
            namespace[name] = ref_cls.__dict__.get(name, getattr(ref_cls, name))
    base_opd = GeometricTrace.opd

    def opd(self, *args, **kwargs):
        return tuple(np.asarray(a).view(LegacyArray)
                     for a in base_opd(self, *args, **kwargs))
    opd.__doc__ = base_opd.__doc__
    namespace["opd"] = opd
    if engine_factory is not None:
        def engine(self):
            if self._engine is None:
                self._engine = engine_factory()
            return self._engine
        namespace["engine"] = property(engine)
    cls = type("GeometricTrace", (GeometricTrace,), namespace)
    if install:
        rayopt.geometric_trace.GeometricTrace = cls
        rayopt.GeometricTrace = cls
        analysis = getattr(rayopt, "analysis", None)
        if analysis is not None and hasattr(analysis, "GeometricTrace"):
            analysis.GeometricTrace = cls
    return cls


def restore(rayopt):
    cls = rayopt.geometric_trace.GeometricTrace
    ref = getattr(cls, "_reference_class", None)
    if ref is not None:
        rayopt.geometric_trace.GeometricTrace = ref
        rayopt.GeometricTrace = ref
        analysis = getattr(rayopt, "analysis", None)
        if analysis is not None and hasattr(analysis, "GeometricTrace"):
            analysis.GeometricTrace = ref
