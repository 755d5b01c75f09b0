"""Generated batches (fields x pupil grid): stand-alone generation kernel +
trace that reads row 0, against the first trace building the rays itself."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P


def main():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    fields = np.c_[np.zeros(5), [0., .35, .5, .7, 1.]]
    rng = np.random.default_rng(0)
    r, phi = .9*np.sqrt(rng.random(2_000_000)), 2*np.pi*rng.random(2_000_000)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    from rayopt_amd.aiming import entrance_pupil
    z, a = entrance_pupil(system)
    g = ra.GeometricTrace(system)          # clocks settle before anything
    g.rays_fields(fields, yp, z, a)        # is compared (~50 launches)
    for rep in range(150):
        g.propagate(clip=True)
    for fuse in (0, 1, 0, 1, 0, 1):
        g = ra.GeometricTrace(system)
        g.engine.set_option("fuse_generate", fuse)
        tot, gen = [], []
        for rep in range(30):
            g.rays_fields(fields, yp, z, a)
            gms = g.kernel_ms()
            g.propagate(clip=True)
            tot.append(gms + g.kernel_ms())
            gen.append(gms)
        print("fuse_generate=%d  generation %.3f ms + trace %.3f ms = %.3f ms "
              "per 10^7 generated rays x 12 surfaces" % (
                  fuse, np.median(gen[-10:]),
                  np.median(tot[-10:]) - np.median(gen[-10:]),
                  np.median(tot[-10:])))


if __name__ == "__main__":
    main()
