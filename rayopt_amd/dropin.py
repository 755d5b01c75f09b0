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
from .geometric_trace import GeometricTrace

BORROWED = ("rays", "rays_point", "rays_line", "rays_clipping",
            "rays_paraxial", "resize", "plot", "print_trace")


def accelerate(rayopt, install=True, engine_factory=None):
    ref_cls = rayopt.geometric_trace.GeometricTrace
    if getattr(ref_cls, "_mi355", False):
        return ref_cls
    namespace = {"_mi355": True, "_reference_class": ref_cls,
                 "__doc__": GeometricTrace.__doc__}
    for name in BORROWED:
        if hasattr(ref_cls, name):
            namespace[name] = ref_cls.__dict__.get(name, getattr(ref_cls, name))
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
