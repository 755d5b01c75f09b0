"""C4 (BASELINE configs[3]: six even aspheres, 10^7 rays, field 17.5 deg),
`reps` launches of the trace kernel with fast_asphere = argv[1]; for
rocprofv3."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P

fast = int(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
system = ra.system_from_yaml(P.ASPHERE_PHONE)
deg = 17.5
y, u = ra.bundles.disc_bundle(10**7, 0.6, deg, 3)
y[:, 1] -= 0.5*np.tan(np.radians(deg))
g = ra.GeometricTrace(system)
g.engine.set_option("fast_asphere", fast)
g.rays_given(y, u)
ms = []
for _ in range(reps):
    g.propagate(clip=True)
    ms.append(g.kernel_ms())
print("fast_asphere=%d: median of last 10 launches %.4f ms" % (
    fast, float(np.median(ms[-10:]))))
