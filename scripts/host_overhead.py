"""Launch-bound regime: wall time of one GeometricTrace.propagate() for small
batches (aiming iterations, merit evaluations), where the kernel takes a few
microseconds and the host path decides."""
import cProfile
import os
import pstats
import sys
import time


sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P


def main():
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    for n in (64, 10_000, 1_000_000):
        y, u = ra.bundles.disc_bundle(n, 17., 5., 1, P.DOUBLE_GAUSS_PUPIL_Z)
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        for _ in range(20):
            g.propagate(clip=True)
        g.engine.sync()
        reps = 300
        t0 = time.perf_counter()
        for _ in range(reps):
            g.propagate(clip=True)
        g.engine.sync()
        dt = (time.perf_counter() - t0)/reps
        t0 = time.perf_counter()
        for _ in range(reps):
            g.propagate(clip=True, keep=[-1])
            r = g.rms()
        dt2 = (time.perf_counter() - t0)/reps
        print("n=%8d  propagate %.1f us (kernel %.1f us)   "
              "propagate(keep=[-1]) + rms() %.1f us" % (
                  n, dt*1e6, g.kernel_ms()*1e3, dt2*1e6))
    y, u = ra.bundles.disc_bundle(10_000, 17., 5., 1, P.DOUBLE_GAUSS_PUPIL_Z)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(300):
        g.propagate(clip=True, keep=[-1])
        g.rms()
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)


if __name__ == "__main__":
    main()
