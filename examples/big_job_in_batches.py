"""A job traced as several batches: 3*10^7 rays (five field bundles of a
double Gauss) as batches of 10^7 rays, one GeometricTrace each -- e.g. because
the rays arrive in portions, or to keep a batch's host arrays small.  Each
batch is an ordinary trace; the per-field statistics of the job are the
ray-count-weighted combination of the batches' (parallel-axis theorem).

(Speed is no reason any more: the engine lays a big batch out in blocks
itself, csrc/rt_lay.h -- 10^8 rays as one batch 10.4 ms per trace, as ten
batches traced in turn 10.4 ms.  Before that layout one batch took 12.0 ms.)

    python examples/big_job_in_batches.py [total_rays] [rays_per_batch]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P

FIELDS = np.c_[np.zeros(5), (0., .35, .5, .7, 1.)]


def pupil_points(n, seed):
    """n points uniform on the unit disc."""
    rng = np.random.default_rng(seed)
    r, phi = np.sqrt(rng.random(n)), 2*np.pi*rng.random(n)
    return np.c_[r*np.cos(phi), r*np.sin(phi)]


def combine(stats):
    """Per-field (count, centroid, rms) of the whole job from the batches'
    spot statistics (count, cx, cy, mean d^2 about the batch centroid, ...):
    the parallel-axis theorem, exact up to rounding."""
    cnt = sum(s[:, 0] for s in stats)
    c = sum(s[:, 0, None]*s[:, 1:3] for s in stats)/cnt[:, None]
    var = sum(s[:, 0]*(s[:, 3] + np.square(s[:, 1:3] - c).sum(1))
              for s in stats)/cnt
    return cnt, c, np.sqrt(var)


def main(total=30_000_000, batch=10_000_000, verbose=True, device=None):
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    nf = len(FIELDS)
    k = -(-total//batch)
    per_field = total//k//nf//64*64          # rays_fields: F*P a multiple of 64
    z = np.full(nf, P.DOUBLE_GAUSS_PUPIL_Z)
    traces = []
    for b in range(k):
        g = ra.GeometricTrace(system, device=device)
        g.rays_fields(FIELDS, pupil_points(per_field, b), z, 17.)
        traces.append(g)
    t0 = time.perf_counter()
    for g in traces:                          # launches queue up ...
        g.propagate(clip=True, keep=[-1])
    for g in traces:                          # ... and finish here
        g.engine.sync()
    dt = time.perf_counter() - t0
    stats = [g.spot_stats(group_rays=per_field) for g in traces]
    cnt, centroid, rms = combine(stats)
    if verbose:
        rays = k*nf*per_field
        print("%d rays as %d batches of %d: %.2f ms (first trace of each "
              "batch, rays built inside it), %.3g ray-surface-ops/s"
              % (rays, k, nf*per_field, dt*1e3,
                 rays*(len(system) - 1)/dt))
        for f in range(nf):
            print("  field %.2f: %9d rays arrived, centroid (%+.5f, %+.5f), "
                  "rms %.5f" % (FIELDS[f, 1], cnt[f], *centroid[f], rms[f]))
    return cnt, centroid, rms, traces


if __name__ == "__main__":
    args = [int(float(a)) for a in sys.argv[1:3]]
    main(*args)
