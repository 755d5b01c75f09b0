"""Polychromatic spot report: for every wavelength, all field points are aimed
(batched, GPU), their ray bundles built on the GPU, traced, and the per-field
RMS spot radii computed -- the numbers rayopt's Analysis prints per field,
for hundreds of fields x 10^4..10^6 rays at once.

    python examples/spot_report.py [rays_per_field]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra


def spot_report(system, fields, nrays=1000, distribution="hexapolar",
                wavelengths=None):
    fields = np.atleast_2d(fields)
    rows = []
    for l in wavelengths or system.wavelengths:
        g = ra.GeometricTrace(system)
        g.rays_points(fields, wavelength=l, nrays=nrays,
                      distribution=distribution)
        P = g.rays_per_field
        xy = np.asarray(g.y[-1])[:, :2].reshape(len(fields), P, 2)
        centroid = np.nanmean(xy, axis=1)
        rms = np.sqrt(np.nanmean(np.square(xy - centroid[:, None]).sum(2), 1))
        rows.append((l, centroid, rms, g.kernel_ms()))
    return rows


def main():
    nrays = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    indices = ra.prescriptions.COOKE_INDICES
    fields = np.c_[np.zeros(11), np.linspace(0, 1, 11)]
    t0 = time.perf_counter()
    out = []
    for l in sorted(indices):
        system = ra.system_from_yaml(ra.prescriptions.cooke(l))
        out += spot_report(system, fields, nrays, wavelengths=[l])
    dt = time.perf_counter() - t0
    print("Cooke triplet, %d fields x %d wavelengths, ~%d rays each: %.2f s"
          % (len(fields), len(out), nrays, dt))
    print(" field " + "".join("  %7.1f nm" % (l*1e9) for l, *_ in out))
    for f, yo in enumerate(fields):
        print(" %5.2f " % yo[1] + "".join("  %10.5f" % r[2][f] for r in out))


if __name__ == "__main__":
    main()
