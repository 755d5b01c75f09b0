"""Placement statistics of ONE box (round 5): several contexts per shape --
C2 (Cooke, 3 x 10^6 rays, one launch), the headline C3 (double-Gauss, 10^7
rays, collimated bundles) and C3' (the same with per-ray launch directions)
-- each with what rt_placement measured (classes, sets of pieces tried, GB/s of the
store pattern over each) and the settled launch time of
its trace.  One JSON line per context; run on several fresh boxes, the table
goes to profiles/r05_final/boxstat/.

    python scripts/boxstat.py [contexts per shape, default 5] [placement 0/1]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import rayopt_amd as ra                                    # noqa: E402
from rayopt_amd import prescriptions as P                  # noqa: E402
import digest_cases as dc                                  # noqa: E402
from bench import workload_rays                            # noqa: E402


def settled_ms(g, clip=True, settle_s=.3, dwell_s=.4):
    eng = g.engine
    g.propagate(clip=clip)
    eng.sync()
    per = max(1, min(10, int(40./max(g.kernel_ms(), 1e-3))))
    t_end = time.perf_counter() + settle_s
    while time.perf_counter() < t_end:
        for _ in range(per):
            eng.trace(1, 0, clip)
        eng.sync()
    t = []
    t_end = time.perf_counter() + dwell_s
    while time.perf_counter() < t_end or len(t) < 5:
        eng.event_record(0)
        for _ in range(per):
            eng.trace(1, 0, clip)
        eng.event_record(1)
        t.append(eng.event_elapsed(0, 1)/per)
    return float(np.median(t))


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    placed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    y2, u2 = dc.bundle(10**6, 5.5, 5., 0)
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y3, u3 = workload_rays(10_000_000, 0)
    y3p, u3p = workload_rays(10_000_000, 7)
    rng = np.random.default_rng(3)
    u3p[:, 0] += 1e-7*rng.standard_normal(len(u3p))
    u3p[:, 1] += 1e-7*rng.standard_normal(len(u3p))
    u3p[:, 2] = np.sqrt(1. - u3p[:, 0]**2 - u3p[:, 1]**2)
    shapes = (("C2", s2, y2, u2, ls), ("C3", s3, y3, u3, None),
              ("C3'", s3, y3p, u3p, None))
    for k in range(reps):
        for name, system, y, u, l in shapes:
            eng = ra.Engine()
            eng.set_option("placement", placed)
            g = ra.GeometricTrace(system, engine=eng)
            t0 = time.perf_counter()
            if l is None:
                g.rays_given(y, u)
            else:
                g.rays_given(y, u, l=l)
            seed_s = time.perf_counter() - t0
            ms = settled_ms(g)
            pl = eng.placement()
            print(json.dumps({"shape": name, "context": k, "trace_ms": ms,
                              "rays_given_s": seed_s, "placement": pl,
                              "blocks": eng.blocks()[0]}), flush=True)
            eng.close()
            del g


if __name__ == "__main__":
    main()
