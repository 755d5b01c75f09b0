"""Generate tests/golden/*.npz by running the UNMODIFIED reference.

Run in the build container (needs /root/reference):

    python tests/golden/make_golden.py

For every case of ``cases.py`` the reference's own
``GeometricTrace.rays_given`` + ``propagate`` (rayopt/geometric_trace.py:49-80)
is executed and its ``y,u,i,t,n`` arrays stored together with the inputs.
Additionally the reference's only numeric known-answer test on this path,
``test_quadrature`` (rayopt/test/test_raytrace.py:189-199: rms == 0.052 for
13 Radau rays at field (0,1) of the Cooke fixture), is reproduced on the
reference's own fixture (catalogue glasses) and stored with its launch rays
and per-surface indices as ``kat_cooke_quadrature.npz``.
"""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))       # tests/: random_systems

from oracle import refshim  # noqa: E402
from cases import cases  # noqa: E402


def run_reference(ro, case):
    s = ro.system_from_yaml(case["yaml"])
    g = ro.GeometricTrace(s)
    g.rays_given(case["y"], case["u"], case["l"])
    # rows outside [start, stop) stay np.empty in the reference: poison them
    # so the tests only look at rows that were traced
    g.y[1:] = np.nan
    g.u[1:] = np.nan
    g.i[1:] = np.nan
    g.t[1:] = np.nan
    g.n[1:] = np.nan
    with np.errstate(all="ignore"):
        g.propagate(start=case["start"], stop=case["stop"], clip=case["clip"])
    return g


def main():
    ro = refshim.load()
    import scipy
    meta = "rayopt@/root/reference numpy %s scipy %s" % (np.__version__,
                                                         scipy.__version__)
    only = sys.argv[1:]          # name prefixes: regenerate just those cases
    for case in cases():
        if only and not case["name"].startswith(tuple(only)):
            continue
        g = run_reference(ro, case)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(
            path, yaml=case["yaml"], y0=g.y[0], u0=g.u[0], l=case["l"],
            clip=case["clip"], start=case["start"],
            stop=-999 if case["stop"] is None else case["stop"],
            y=g.y, u=g.u, i=g.i, t=g.t, n=g.n, meta=meta)
        print("%-28s L=%d N=%d nan(u)=%.3f" % (
            case["name"], g.y.shape[0], g.y.shape[1],
            np.isnan(g.u[1:, :, 0]).mean()))

    if only:
        return
    # --- consumers: rms / refocus / opd(resample=0) from the reference ---
    from cases import consumer_cases
    for case in consumer_cases():
        s = ro.system_from_yaml(case["yaml"])
        if case.get("finite_object"):
            s.object = ro.FiniteConjugate(radius=1.)
        g = ro.GeometricTrace(s)
        g.rays_given(case["y"], case["u"], case["l"], case["w"], case["ref"])
        with np.errstate(all="ignore"):
            g.propagate(clip=case["clip"])
            out = dict(rms_mean=g.rms(), rms_ref=g.rms(ref=case["ref"]),
                       rms_mid=g.rms(i=2))
            x, y, t = g.opd(radius=case["radius"], resample=0)
            d0 = float(s[-1].distance)
            g.refocus()
            out["refocus_shift"] = float(s[-1].distance) - d0
            out["y_after_refocus"] = g.y[-1].copy()
        np.savez_compressed(
            os.path.join(HERE, "consumers_" + case["name"] + ".npz"),
            yaml=case["yaml"], y0=case["y"], u0=case["u"], l=case["l"],
            w=case["w"] if case["w"] is not None else np.zeros(0),
            ref=case["ref"], clip=case["clip"], radius=case["radius"],
            finite_object=bool(case.get("finite_object")),
            opd_x=x, opd_y=y, opd_t=t, meta=meta, **out)
        print("consumers_%-20s rms=%.6g shift=%.6g opd rms=%.4g waves" % (
            case["name"], out["rms_mean"], out["refocus_shift"],
            np.nanstd(t)))

    # --- the reference's own known-answer test -------------------------
    sys.path.insert(0, refshim.REFERENCE_ROOT)
    db = os.path.join(tempfile.mkdtemp(), "library.sqlite")
    shutil.copy(os.path.join(refshim.REFERENCE_ROOT, "rayopt",
                             "library.sqlite"), db)
    ro.library.Library.one(db="sqlite:///" + db)
    from rayopt.test.test_raytrace import cooke
    s = ro.system_from_yaml(cooke)
    s.update()
    p = ro.ParaxialTrace(s)
    p.update_conjugates()
    g = ro.GeometricTrace(s)
    g.rays_point((0, 1.), nrays=13, distribution="radau", filter=False)
    rms = g.rms()
    assert abs(rms - .052) < .052e-2*10, rms
    geom = [dict(distance=float(e.distance), curvature=float(e.curvature),
                 radius=float(e.radius)) for e in s]
    np.savez_compressed(
        os.path.join(HERE, "kat_cooke_quadrature.npz"),
        y0=g.y[0], u0=g.u[0], w=g.w, ref=g.ref, l=g.l, n=g.n, y=g.y, u=g.u,
        i=g.i, t=g.t, rms=rms,
        distance=[d["distance"] for d in geom],
        curvature=[d["curvature"] for d in geom],
        radius=[d["radius"] for d in geom], meta=meta)
    print("kat_cooke_quadrature rms=%.16g (pin 0.052 rtol 1e-2)" % rms)
    shutil.rmtree(os.path.dirname(db), ignore_errors=True)


if __name__ == "__main__":
    main()
