"""Calibrate the memory-system ceiling for the trace kernel's access pattern
on the headline workload (C3, 10^7 rays): store pattern without arithmetic,
linear fill, copy -- next to the trace kernel itself."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    eng = g.engine
    res = {}
    for rep in range(3):
        g.propagate(clip=True)
    ms = []
    for rep in range(10):
        g.propagate(clip=True)
        ms.append(eng.kernel_ms())
    alg = n*(56*12 + 48)
    res["trace_alias_i"] = dict(ms=float(np.median(ms)),
                                GBs=alg/np.median(ms)/1e6)
    eng.set_option("alias_i", 0)
    ms = []
    for rep in range(10):
        g.propagate(clip=True)
        ms.append(eng.kernel_ms())
    alg = n*(80*12 + 48)
    res["trace_full_i"] = dict(ms=float(np.median(ms)),
                               GBs=alg/np.median(ms)/1e6)
    for mode, name in ((0, "store_pattern_80B"), (1, "linear_fill"),
                       (3, "fill_one_store_per_lane"),
                       (4, "fill_one_store_per_lane_nt"), (2, "copy")):
        t = []
        for rep in range(8):
            m, b = eng.probe(mode)
            t.append(m)
        res[name] = dict(ms=float(np.median(t[2:])), GBs=b/np.median(t[2:])/1e6)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
