"""Wall time of batched aiming and of one polychromatic SpotOperand
evaluation (aiming + generation + trace + grouped reduction)."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import merit
from rayopt_amd.aiming import FieldAimer


def main():
    system = ra.system_from_yaml(ra.prescriptions.COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    for nf in (3, 200):
        fields = np.c_[np.zeros(nf), np.linspace(0, 1, nf)]
        aimer = FieldAimer(system, engine=ra.get_engine())
        aimer.pupil(fields)
        t0 = time.perf_counter()
        for _ in range(5):
            aimer.pupil(fields)
        print("pupil(%d fields): %.2f ms" % (nf, (time.perf_counter() - t0)/5*1e3))
    op = merit.SpotOperand(system, np.c_[np.zeros(3), [0., .7, 1.]],
                           nrays=600, distribution="hexapolar", clip=False,
                           weight=1.)
    op.get()
    t0 = time.perf_counter()
    for _ in range(5):
        op.get()
    print("SpotOperand.get (3 wavelengths x 3 fields, aimed): %.2f ms, "
          "kernel %.3f ms" % ((time.perf_counter() - t0)/5*1e3,
                              op.kernel_ms[-1]))
    op.aim = False
    op.get()
    t0 = time.perf_counter()
    for _ in range(5):
        op.get()
    print("same, aim=False: %.2f ms" % ((time.perf_counter() - t0)/5*1e3))
    pr = cProfile.Profile()
    pr.enable()
    aimer.pupil(np.c_[np.zeros(3), [0., .7, 1.]])
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)


if __name__ == "__main__":
    main()
