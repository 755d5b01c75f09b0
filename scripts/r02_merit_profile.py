"""Where one merit evaluation spends its time (round 2): wall time of
rt_aim_pupil by number of fields, of SpotOperand.get aimed / unaimed on a
system that changes between calls (as under an optimiser), and the host
profile of both.  Prints JSON lines, then the cProfile tables."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import merit
from rayopt_amd.aiming import FieldAimer


def wall(fn, reps=20):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return float(np.median(t))*1e3


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else ""
    cooke = ra.system_from_yaml(ra.prescriptions.COOKE % dict(
        air=1.0, sk16="1.62041/60.32", f2="1.62004/36.37"))
    dgauss = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    for name, system in (("cooke", cooke), ("double_gauss", dgauss)):
        for nf in (1, 3, 9, 200, 2000, 20000):
            fields = np.c_[np.zeros(nf), np.linspace(0, 1, nf)]
            aimer = FieldAimer(system, engine=ra.get_engine())
            ms = wall(lambda: aimer.pupil(fields), 10)
            print(json.dumps(dict(tag=tag, what="FieldAimer.pupil",
                                  system=name, fields=nf, wall_ms=ms)),
                  flush=True)
    fields = np.c_[np.zeros(3), [0., .7, 1.]]
    out = io.StringIO()
    for aim in (True, False):
        op = merit.SpotOperand(cooke, fields, nrays=600,
                               distribution="hexapolar", clip=False,
                               weight=1., aim=aim)

        def step():
            cooke[2].curvature *= 1.0000001
            return op.get()
        ms = wall(step, 40)
        print(json.dumps(dict(tag=tag, what="SpotOperand.get, 3 wavelengths "
                              "x 3 fields x 600 rays, changing system",
                              aim=aim, wall_ms=ms,
                              kernel_ms=op.kernel_ms[-1])), flush=True)
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(50):
            step()
        pr.disable()
        out.write("\n==== aim=%s, 50 evaluations ====\n" % aim)
        pstats.Stats(pr, stream=out).sort_stats("tottime").print_stats(28)
    print(out.getvalue())


if __name__ == "__main__":
    main()
