"""Re-traces of a device-generated batch: launch rays rebuilt in registers
(regenerate=1, default) against read from row 0 (regenerate=0).  Same
process, alternating, after the clocks have settled."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.aiming import entrance_pupil


def main():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    fields = np.c_[np.zeros(5), [0., .35, .5, .7, 1.]]
    rng = np.random.default_rng(0)
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    r, phi = .9*np.sqrt(rng.random(m)), 2*np.pi*rng.random(m)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    z, a = entrance_pupil(system)
    g = ra.GeometricTrace(system)
    g.rays_fields(fields, yp, z, a)
    for _ in range(300):
        g.propagate(clip=True)
    out = {0: [], 1: []}
    for rnd in range(6):
        for regen in (0, 1):
            g.engine.set_option("regenerate", regen)
            ms = []
            for _ in range(8):
                g.propagate(clip=True)
                ms.append(g.kernel_ms())
            out[regen].append(float(np.median(ms)))
    for regen in (0, 1):
        print("regenerate=%d  %s  best %.4f ms (%d rays x 12 surfaces)" % (
            regen, " ".join("%.4f" % t for t in out[regen]),
            min(out[regen]), 5*m))


if __name__ == "__main__":
    main()
