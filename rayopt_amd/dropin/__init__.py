"""Swap the MI355X engine into an installed rayopt.

``accelerate(rayopt)`` returns (and installs) a ``GeometricTrace`` class whose
storage and ``propagate()`` are this package's (device resident, HIP kernel)
while the ray-generation helpers that only *call* ``rays_given``/``propagate``
-- ``rays, rays_point, rays_line, rays_clipping, rays_paraxial, resize, plot,
print_trace`` (rayopt/geometric_trace.py:185-259) -- are taken from the
installed rayopt unchanged, so aiming, pupils and conjugates keep running
rayopt's own host code.  ``rayopt.GeometricTrace`` and the name imported by
``rayopt.analysis`` are rebound, so ``Analysis`` traces on the GPU too; with
``aim=True`` (default) ``System.pupil`` -- ~130 serial one-ray traces per
field in the reference -- is answered by the aiming kernel.
"""
import numpy as np

from ..geometric_trace import GeometricTrace

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


def _device_pupil(reference_pupil, engine_factory):
    """``System.pupil`` (rayopt/system.py:585-593) with the root finding of
    ``_aim_pupil`` done by the aiming kernel: same starting values (the
    object pupil the paraxial trace left in ``system.object.pupil``), same
    return value ``(z, a[2][2])``; the reference's own code handles what the
    kernel does not model (``pupil.aim`` off, an explicit ``stop`` index)."""
    from ..aiming import FieldAimer
    from ..engine import get_engine

    def pupil(self, yo, l=None, stop=None, **kwargs):
        pup = self.object.pupil
        if stop not in (None, -1) or not pup.aim or kwargs:
            return reference_pupil(self, yo, l=l, stop=stop, **kwargs)
        # kept inside the reference's own cache, under a key of its own, so
        # that System.update() -- which clears _pupil_cache after any change
        # of the prescription (rayopt/system.py:201-202) -- forgets these
        # results together with the reference's
        cache = self._pupil_cache.setdefault(("mi355", l, stop), {})
        key = (float(yo[0]), float(yo[1]), float(pup.distance),
               float(pup.radius), len(self))
        if key not in cache:
            engine = engine_factory() if engine_factory else get_engine()
            aimer = FieldAimer(self, self.wavelengths[0] if l is None else l,
                               engine=engine)
            z, a = aimer.pupil([yo], float(pup.distance), float(pup.radius),
                               rim=(stop == -1))
            cache[key] = (float(z[0]), a[0])
        z, a = cache[key]
        return z, a.copy()
    pupil._mi355 = True
    pupil._reference = reference_pupil
    return pupil


def accelerate(rayopt, install=True, engine_factory=None, aim=True):
    if install and aim and not getattr(rayopt.system.System.pupil, "_mi355",
                                       False):
        rayopt.system.System.pupil = _device_pupil(
            rayopt.system.System.pupil, engine_factory)
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
    pupil = rayopt.system.System.pupil
    if getattr(pupil, "_mi355", False):
        rayopt.system.System.pupil = pupil._reference
    cls = rayopt.geometric_trace.GeometricTrace
    ref = getattr(cls, "_reference_class", None)
    if ref is not None:
        rayopt.geometric_trace.GeometricTrace = ref
        rayopt.GeometricTrace = ref
        analysis = getattr(rayopt, "analysis", None)
        if analysis is not None and hasattr(analysis, "GeometricTrace"):
            analysis.GeometricTrace = ref
