"""Wall-clock of the whole user-visible call sequence for one spot diagram:
host rays -> rays_given (PCIe) -> propagate -> image intercepts back on the
host, against the numpy port of the reference's path on one core (sampled)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from bench import workload_rays


def main(n=10_000_000):
    system = ra.system_from_yaml(ra.prescriptions.DOUBLE_GAUSS)
    y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y[:1000], u[:1000])      # context + code pages warm
    g.propagate(clip=True)
    times = []
    for rep in range(4):
        t0 = time.perf_counter()
        g.rays_given(y, u)
        t1 = time.perf_counter()
        g.propagate(clip=True)
        g.engine.sync()
        t2 = time.perf_counter()
        spot = g.y[-1, :, :2]
        t3 = time.perf_counter()
        r = g.rms()
        t4 = time.perf_counter()
        times.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0))
    up, tr, down, rms, tot = np.array(times[1:]).min(0)
    print("%d rays x %d surfaces: upload %.1f ms, trace %.1f ms, image row "
          "to host %.1f ms, device rms %.2f ms: %.1f ms end to end "
          "(%.2e ray-surface-ops/s incl. PCIe)" % (
              n, len(system) - 1, up*1e3, tr*1e3, down*1e3, rms*1e3, tot*1e3,
              n*(len(system) - 1)/tot))
    return spot, r


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000)
