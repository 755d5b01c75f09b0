"""TEST INFRASTRUCTURE ONLY -- the reference's own GeometricTrace.propagate
timed on this host, for bench.py's ``cpu_baseline`` leg (``kind:
"reference"``) and scripts/cpu_reference_here.py.

``/root/reference`` exists in the build container and not on the GPU pool, so
on a GPU box :func:`time_reference` returns None and the bench reports the
numpy port (``kind: "port"``) instead; both are timed side by side wherever
the reference is present, which is how "the port runs the reference's own
numpy operations at the reference's speed" is checked rather than asserted.
"""
import os
import time

import numpy as np

from . import refshim


def time_reference(y, u, wavelength, clip, want_image_row=None,
                   max_rays=2_000_000, prescription=None):
    """One ``rayopt.GeometricTrace.propagate(clip=clip)``
    (rayopt/geometric_trace.py:72-80) of the first ``max_rays`` rays through
    the double-Gauss of BASELINE configs[2], one core.  None if the reference
    tree is not on this box."""
    if not refshim.available():
        return None
    from rayopt_amd import prescriptions as P
    ro = refshim.load()
    import yaml
    text = P.DOUBLE_GAUSS if prescription is None else prescription
    system = ro.system_from_yaml(text)
    m = min(len(y), max_rays)
    trace = ro.GeometricTrace(system)
    warm = min(100000, max(1, m//10))
    trace.rays_given(y[:warm], u[:warm], wavelength)
    with np.errstate(all="ignore"):
        trace.propagate(clip=clip)                  # warm
    trace = ro.GeometricTrace(system)
    trace.rays_given(y[:m], u[:m], wavelength)
    t0 = time.perf_counter()
    with np.errstate(all="ignore"):
        trace.propagate(clip=clip)
    dt = time.perf_counter() - t0
    S = len(system) - 1
    same = None
    if want_image_row is not None:
        same = bool(np.array_equal(trace.y[-1], want_image_row[:m],
                                   equal_nan=True))
    return {
        "value": m*S/dt, "unit": "ray-surface-ops/s", "cores": 1,
        "kind": "reference",
        "sample": "first %d rays of the same workload, one "
                  "rayopt.GeometricTrace.propagate() imported in place from "
                  "%s (%.1f s); image row bit-identical to the numpy port: "
                  "%s; host has %d cores" % (m, refshim.REFERENCE_ROOT, dt,
                                             same, os.cpu_count()),
    }
