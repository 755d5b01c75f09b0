"""fast_asphere vs the exact path on the device, every array: how many values
differ, by how much (C4, 2*10^5 rays), and the NaN masks."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P

system = ra.system_from_yaml(P.ASPHERE_PHONE)
n = 200_000
for deg in (0., 17.5):
    y, u = ra.bundles.disc_bundle(n, 0.6, deg, 3)
    y[:, 1] -= 0.5*np.tan(np.radians(deg))
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    g.engine.set_option("fast_asphere", 0)
    g.propagate(clip=True)
    exact = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
    ms0 = g.kernel_ms()
    g.engine.set_option("fast_asphere", 1)
    g.propagate(clip=True)
    fast = {k: np.array(np.asarray(getattr(g, k))) for k in "yuit"}
    ms1 = g.kernel_ms()
    out = {"field_deg": deg, "rays": n, "kernel_ms_exact": ms0,
           "kernel_ms_fast": ms1}
    for k in "yuit":
        a, b = exact[k], fast[k]
        fin = np.isfinite(a) & np.isfinite(b)
        scale = np.abs(a[fin]).max()
        out[k] = {"nan_masks_identical": bool(np.array_equal(np.isnan(a),
                                                             np.isnan(b))),
                  "values_differing": int((a[fin] != b[fin]).sum()),
                  "values": int(fin.sum()),
                  "max_abs_dev_over_scale": float(np.abs(a[fin] - b[fin]).max()
                                                  / scale)}
    print(json.dumps(out), flush=True)
    g.engine.close()
