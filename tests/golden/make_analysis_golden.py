"""Record what rayopt's Analysis asks of GeometricTrace, and what it gets.

Run in the build container (needs /root/reference):

    python tests/golden/make_analysis_golden.py

``rayopt.analysis.Analysis.run`` (rayopt/analysis.py:76-143) and the plot
helpers it calls (transverse / longitudinal / spots / opds, :219-410) are
executed unmodified on the UNMODIFIED reference trace, wrapped only so that
every outermost call on a GeometricTrace instance is logged -- method,
arguments -- together with a snapshot of the state Analysis reads afterwards
(``nrays, ref, y[0], u[0], y[-1], i[-1]``; the shift of ``refocus``; the grid
of ``opd``; peak and centroid of ``psf``; ``str(trace)``).  Bundles larger
than 200 rays are stored as every 7th ray.  The result,
``tests/golden/analysis_<name>.npz``, lets the GPU test replay the same call
sequence through ``rayopt_amd.GeometricTrace`` on a box that has no
/root/reference (tests/test_analysis_replay.py).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import matplotlib  # noqa: E402
matplotlib.use("Agg")

from oracle import refshim  # noqa: E402
from rayopt_amd import dropin, prescriptions as P  # noqa: E402

RECORDED = ("rays_paraxial", "rays_point", "rays_clipping", "rays_line",
            "rays_given", "propagate", "refocus", "resize", "opd", "psf",
            "__str__")
STRIDE_ABOVE, STRIDE = 200, 7


def jsonable(v):
    if isinstance(v, np.ndarray):
        return {"array": v.tolist()}
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (tuple, list)):
        return [jsonable(x) for x in v]
    return v


def recording_class(base, log, arrays):
    state = {"depth": 0, "ids": {}}

    def snapshot(self, key):
        n = self.nrays
        pick = slice(None, None, STRIDE if n > STRIDE_ABOVE else 1)
        arrays[key + "_y0"] = self.y[0][pick].copy()
        arrays[key + "_u0"] = self.u[0][pick].copy()
        arrays[key + "_ylast"] = self.y[-1][pick].copy()
        arrays[key + "_ilast"] = self.i[-1][pick].copy()
        return {"nrays": int(n), "ref": jsonable(self.ref),
                "stride": pick.step or 1,
                "image_distance": float(self.system[-1].distance)}

    def wrap(name):
        inner = getattr(base, name)

        def method(self, *args, **kwargs):
            outer = state["depth"] == 0
            state["depth"] += 1
            try:
                with np.errstate(all="ignore"):
                    out = inner(self, *args, **kwargs)
            finally:
                state["depth"] -= 1
            if not outer:
                return out
            tid = state["ids"].setdefault(id(self), len(state["ids"]))
            k = len(log)
            rec = {"trace": tid, "method": name, "args": jsonable(args),
                   "kwargs": jsonable(kwargs)}
            key = "c%03d" % k
            if name == "__str__":
                rec["text"] = out
            elif name == "opd":
                # numpy 2 dropped ndarray.ptp, which Analysis calls on the
                # result (rayopt/analysis.py:314)
                out = tuple(np.asarray(v).view(dropin.LegacyArray)
                            for v in out)
                x, y, o = out
                arrays[key + "_opd_x"], arrays[key + "_opd_y"] = x, y
                arrays[key + "_opd"] = o
            elif name == "psf":
                x, y, p = out
                rec["psf_shape"] = list(p.shape)
                rec["psf_peak"] = float(p.max())
                rec["psf_step"] = float(x[1, 0] - x[0, 0])
                rec["psf_sum"] = float(p.sum())
                rec["psf_centroid"] = [float((p*x).sum()/p.sum()),
                                       float((p*y).sum()/p.sum())]
            if name not in ("__str__", "opd", "psf", "resize") and \
                    hasattr(self, "y") and self.nrays:
                rec.update(snapshot(self, key))
            log.append(rec)
            return out
        method.__name__ = name
        return method

    return type("RecordingTrace", (base,),
                {name: wrap(name) for name in RECORDED})


def record(name, text, **options):
    ro = refshim.load()
    dropin.modernize(ro)
    import matplotlib.pyplot as plt
    import rayopt.analysis as analysis
    log, arrays = [], {}
    base = ro.geometric_trace.GeometricTrace
    analysis.GeometricTrace = recording_class(base, log, arrays)
    try:
        system = ro.system_from_yaml(text)
        a = analysis.Analysis(system, print=False, print_full=True,
                              **options)
    finally:
        analysis.GeometricTrace = base
        plt.close("all")
    import scipy
    np.savez_compressed(
        os.path.join(HERE, "analysis_%s.npz" % name), yaml=text,
        calls=json.dumps(log),
        fields=np.asarray(system.fields, dtype=float),
        wavelengths=np.asarray(system.wavelengths, dtype=float),
        meta="rayopt@/root/reference Analysis.run, numpy %s scipy %s "
             "matplotlib %s" % (np.__version__, scipy.__version__,
                                matplotlib.__version__),
        **arrays)
    kinds = {}
    for rec in log:
        kinds[rec["method"]] = kinds.get(rec["method"], 0) + 1
    print(name, len(log), "calls on", len({r["trace"] for r in log}),
          "traces:", kinds, "figures:", len(a.figures))


def main():
    # no aim flag: launches are first-order data -> tight replay tolerance
    record("double_gauss", P.DOUBLE_GAUSS)
    # aim: True, three wavelengths, catalogue-free indices
    record("cooke", P.COOKE % dict(air=1.0, sk16="1.62041/60.32",
                                   f2="1.62004/36.37"))
    # six even aspheres (Newton intercepts), aimed
    record("asphere_phone", P.ASPHERE_PHONE.replace(
        "pupil: {radius: 0.6}", "pupil: {radius: 0.6, aim: True}"))


if __name__ == "__main__":
    main()
