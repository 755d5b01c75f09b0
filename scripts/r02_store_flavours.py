"""Store flavours of the trace kernel's store pattern (no arithmetic): plain,
non-temporal, sc1 (write-through), sc0 sc1 -- rt_probe modes 8/7/6/0 on C3,
10^7 rays."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
eng = g.engine
for _ in range(60):
    g.propagate(clip=True)
names = {0: "plain", 1: "non-temporal", 2: "sc1", 3: "sc0 sc1"}
modes = {8: "56 B/op 8-byte stores, no read", 7: "56 B/op, input read",
         6: "80 B/op 16-byte stores, no read", 0: "80 B/op, input read"}
for rep in range(2):
    for fl in (0, 1, 2, 3):
        eng.set_option("probe_store", fl)
        for mode in (8, 7, 6, 0):
            t = []
            for _ in range(8):
                ms, b = eng.probe(mode)
                t.append(ms)
            ms = float(np.median(t[2:]))
            print(json.dumps(dict(rep=rep, store=names[fl], pattern=modes[mode],
                                  ms=ms, GBs=b/ms/1e6)), flush=True)
eng.set_option("probe_store", 0)
