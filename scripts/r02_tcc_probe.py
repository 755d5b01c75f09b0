"""The store pattern with and without the input read, and the single-stream
fill, `reps` launches each (rt_probe modes 7 / 8 / 3) -- to be run under
rocprofv3 --pmc with the L2 <-> fabric (TCC_EA0_*) counters."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(10_000_000, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
for _ in range(4):
    g.propagate(clip=True)
for mode in (7, 8, 3):
    ms = [g.engine.probe(mode)[0] for _ in range(reps)]
    print("mode %d median %.4f ms" % (mode, float(np.median(ms))))
