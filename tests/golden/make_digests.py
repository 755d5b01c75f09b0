"""Generate tests/golden/digests.json by running the UNMODIFIED reference on
the full-size cases of digest_cases.py (build container, /root/reference):

    python tests/golden/make_digests.py

For every case the reference's GeometricTrace.rays_given + propagate
(rayopt/geometric_trace.py:49-80) is run and SHA-256 digests of its inputs
and of every value of y, u, i, t of the traced rows are stored (NaNs
canonicalised, see digest_cases.canonical_nan).  The asphere case takes the
reference ~20 s (one scipy.optimize.newton call per ray and surface)."""
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import refshim  # noqa: E402
import digest_cases as dc  # noqa: E402


def main():
    ro = refshim.load()
    import scipy
    out = {"_made_with": "rayopt@/root/reference numpy %s scipy %s" % (
        np.__version__, scipy.__version__)}
    only = sys.argv[1:]
    path = os.path.join(HERE, "digests.json")
    if only and os.path.exists(path):
        with open(path) as f:
            out = json.load(f)
    for case in dc.cases():
        if only and case["name"] not in only:
            continue
        y, u = case["rays"]()
        s = ro.system_from_yaml(case["yaml"])
        g = ro.GeometricTrace(s)
        g.rays_given(y, u, case["l"])
        t0 = time.perf_counter()
        with np.errstate(all="ignore"):
            g.propagate(clip=case["clip"])
        dt = time.perf_counter() - t0
        arrays = {"y": g.y, "u": g.u, "i": g.i, "t": g.t}
        L = len(s)

        def rows_of(k, j):
            if j >= L:
                raise IndexError
            return arrays[k][j]
        out[case["name"]] = {
            "rays": int(y.shape[0]), "elements": L, "clip": case["clip"],
            "inputs": dc.digest_inputs(y, u),
            "results": dc.digest_rows(rows_of),
            "dead_at_image": int(np.isnan(g.u[-1][:, 0]).sum()),
            "reference_seconds": round(dt, 2),
        }
        print(case["name"], out[case["name"]], flush=True)
    gen = dc.GENERATED
    if not only or gen["name"] in only:
        s = ro.system_from_yaml(gen["yaml"])
        yp = gen["pupil"]()
        a = gen["a"]*np.array(((-1., -1.), (1., 1.)))
        ys, us = zip(*[s.aim(np.array(yo), yp, gen["z"], a, filter=False)
                       for yo in gen["fields"]])
        y, u = np.concatenate(ys), np.concatenate(us)
        g = ro.GeometricTrace(s)
        g.rays_given(y, u, gen["l"])
        with np.errstate(all="ignore"):
            g.propagate(clip=gen["clip"])
        arrays = {"y": g.y, "u": g.u, "i": g.i, "t": g.t}
        L = len(s)

        def rows_of(k, j):
            if j >= L:
                raise IndexError
            return arrays[k][j]
        out[gen["name"]] = {
            "rays": int(y.shape[0]), "elements": L, "clip": gen["clip"],
            "inputs": dc.digest_inputs(yp, np.array(gen["fields"])),
            "launch": dc.digest_inputs(g.y[0], g.u[0]),
            "results": dc.digest_rows(rows_of),
            "dead_at_image": int(np.isnan(g.u[-1][:, 0]).sum())}
        print(gen["name"], out[gen["name"]], flush=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
