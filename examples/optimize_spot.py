"""The optimiser as a caller of the fast trace.

Minimise the polychromatic-free RMS spot of the singlet by bending the lens
(both curvatures) and refocusing, with 10^6 rays per merit evaluation: every
evaluation is one fused trace that stores only the image row
(``propagate(keep=[-1])``) plus one on-device reduction (``rms``) -- about a
millisecond of GPU time, two scalars over PCIe.
"""
import os
import sys
import time

import numpy as np
from scipy.optimize import minimize

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra


def main(nrays=1_000_000, verbose=True):
    system = ra.system_from_yaml(ra.prescriptions.SINGLET)
    y, u = ra.bundles.disc_bundle(nrays, 7.5, 0., 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    evals = []

    def merit(x):
        system[1].curvature, system[2].curvature = x[0], x[1]
        system[3].distance = x[2]
        g.propagate(keep=[-1])
        r = g.rms()
        evals.append(g.kernel_ms())
        return r if np.isfinite(r) else 1e3

    x0 = np.array([system[1].curvature, system[2].curvature,
                   system[3].distance])
    f0 = merit(x0)
    t0 = time.perf_counter()
    res = minimize(merit, x0, method="Nelder-Mead",
                   options=dict(xatol=1e-7, fatol=1e-9, maxiter=400))
    dt = time.perf_counter() - t0
    if verbose:
        print("rms %.5f -> %.5f in %d evaluations, %.2f s wall, median "
              "kernel %.3f ms for %d rays x 3 surfaces" % (
                  f0, res.fun, res.nfev, dt, float(np.median(evals)), nrays))
    return f0, res.fun, res.nfev


if __name__ == "__main__":
    main()
