"""What does the 48 B/ray input read cost the store-bound pattern, and does
where does that cost sit?  rt_probe modes 0 (rows in HBM), 5 (L2-resident
window), 6 (no read), interleaved in one session."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays


def main():
    n = 10_000_000
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    eng = g.engine
    for _ in range(30):
        g.propagate(clip=True)
    names = {0: "soa_rows", 5: "l2_window", 6: "no_read"}
    t = {m: [] for m in names}
    for rnd in range(6):
        for m in names:
            ms, b = eng.probe(m)
            t[m].append(ms)
    out = {names[m]: dict(ms=[round(x, 4) for x in v],
                          median=float(np.median(v[1:]))) for m, v in t.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
