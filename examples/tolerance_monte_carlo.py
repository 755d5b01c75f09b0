"""Monte-Carlo tolerancing in one launch: N perturbed copies of the Cooke
triplet (curvatures, thicknesses, element tilts drawn from tolerance bands),
the same bundle of rays through every copy -- one surface table per variant,
``GeometricTrace.rays_variants`` -- and the RMS spot of every copy from one
grouped device reduction.
"""
import copy
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra


def main(nvariants=5000, nrays=640, verbose=True):
    base = ra.system_from_yaml(ra.prescriptions.cooke())
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    variants = []
    for _ in range(nvariants):
        s = copy.deepcopy(base)
        for el in s[1:-1]:
            el.curvature *= 1 + 1e-3*rng.standard_normal()    # 0.1 % radius
            el.distance += 0.02*rng.standard_normal()         # 20 um
        for k in (1, 3, 6):                                   # element tilts
            s[k].angles = tuple(3e-4*rng.standard_normal(2)) + (0.,)
        variants.append(s)
    t1 = time.perf_counter()
    y, u = ra.bundles.disc_bundle(nrays, 4., 10., 1)
    g = ra.GeometricTrace(base)
    g.rays_variants(y, u, variants)
    g.propagate(clip=False, keep=[-1])
    rms = g.rms_fields(lost="omit")
    t2 = time.perf_counter()
    nominal = ra.GeometricTrace(base)
    nominal.rays_given(y, u)
    nominal.propagate()
    if verbose:
        q = np.percentile(rms, (50, 90, 99))
        print("%d variants x %d rays: built in %.2f s, packed + traced + "
              "reduced in %.3f s (kernel %.3f ms)" % (
                  nvariants, nrays, t1 - t0, t2 - t1, g.kernel_ms()))
        print("rms spot at 10 deg: nominal %.4f, median %.4f, 90 %% %.4f, "
              "99 %% %.4f" % (nominal.rms(), *q))
    return rms


if __name__ == "__main__":
    main()
