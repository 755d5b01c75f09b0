"""Polychromatic, multi-field optimisation with rayopt's optimiser front end
(rayopt_amd.merit mirrors rayopt/optimize.py) and the GPU doing the work.

Variables: the image distance and the last two curvatures of the Cooke
triplet.  Merit: RMS spot radius of three field points at three wavelengths.
Every merit evaluation is
  * one aiming kernel (9 chief + 36 marginal root finds, rt_aim_pupil),
  * one fused trace of 9 bundles that keeps only the image row,
  * one grouped reduction (rt_spot_stats),
about a millisecond altogether; 72 doubles cross PCIe.
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import merit


def main(nrays=2000, verbose=True, batched=False):
    system = ra.system_from_yaml(ra.prescriptions.COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    system[-1].distance += 0.4                       # start out of focus
    fields = np.c_[np.zeros(3), [0., .7, 1.]]
    spot = merit.SpotOperand(system, fields, nrays=nrays,
                             distribution="hexapolar", clip=False, weight=1.)
    variables = [
        merit.PathVariable(system, (-1, "distance"), bounds=(40., 46.)),
        merit.PathVariable(system, (6, "curvature"), bounds=(0.002, 0.012)),
        merit.PathVariable(system, (7, "curvature"), bounds=(-0.08, -0.04)),
    ]
    before = spot.get().reshape(3, 3)
    t0 = time.perf_counter()
    # batched: the finite-difference gradient (len(variables) + 1 systems)
    # is traced as variants of the system in one launch
    res = merit.optimize(variables, [spot], options=dict(maxiter=40),
                         **(dict(jac="batched") if batched else {}))
    dt = time.perf_counter() - t0
    res.accept()
    after = spot.get().reshape(3, 3)
    if verbose:
        np.set_printoptions(precision=4, suppress=True)
        print("rms spot [wavelength, field] before:\n%s\nafter:\n%s" % (
            before, after))
        print("%s: %d operand evaluations (%d rays per system) in %.3f s; "
              "x = %s" % ("batched gradients" if batched else "scipy "
                          "finite differences", res.nevaluations,
                          spot.trace.nrays, dt,
                          res.x*[v.scale for v in variables]))
    return before, after, res


if __name__ == "__main__":
    main()
    main(batched=True)
