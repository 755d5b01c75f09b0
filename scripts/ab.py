"""A/B kernel variants on the headline workload inside ONE process (boxes
differ by ~8 %, so only same-session ratios mean anything).  Each variant is
checked bit-identical to the default before it is timed."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

DEFAULTS = dict(rays_per_thread=1, nontemporal=0, xcd_remap=0, block=256,
                lds_pad=0,
                alias_i=1, uniform_fix=0, gate_log2=0, gate_window=1)


def main():
    n = int(os.environ.get("RT_AB_RAYS", 10_000_000))
    variants = [dict(v) for v in json.loads(sys.argv[1])] if len(sys.argv) > 1 \
        else [{}]
    if os.environ.get("RT_AB_CONFIG") == "asphere":
        system = ra.system_from_yaml(P.ASPHERE_PHONE)
        y, u = ra.bundles.disc_bundle(n, 0.6, 17.5, 3)
        y[:, 1] -= 0.5*np.tan(np.radians(17.5))
    else:
        system = ra.system_from_yaml(P.DOUBLE_GAUSS)
        y, u = workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    eng = g.engine
    for _ in range(int(os.environ.get("RT_AB_SETTLE", 300))):
        g.propagate(clip=True)      # clocks settle (~50-300 launches)
    ref = [np.array(np.asarray(r[-1])) for r in (g.y, g.u, g.t)]
    ref_mid = np.array(np.asarray(g.y[5]))
    times = {json.dumps(v): [] for v in variants}
    for rnd in range(int(os.environ.get("RT_AB_ROUNDS", 3))):
        for v in variants:
            opts = dict(DEFAULTS)
            opts.update(v)
            for k, val in opts.items():
                eng.set_option(k, val)
            g.propagate(clip=True)
            if rnd == 0 and not os.environ.get("RT_AB_NOCHECK"):
                for rows, want in zip((g.y, g.u, g.t), ref):
                    assert np.array_equal(np.asarray(rows[-1]), want,
                                          equal_nan=True), v
                assert np.array_equal(np.asarray(g.y[5]), ref_mid,
                                      equal_nan=True), v
            ms = []
            for rep in range(6):
                g.propagate(clip=True)
                ms.append(eng.kernel_ms())
            times[json.dumps(v)].append(float(np.median(ms)))
    for k, t in times.items():
        print("%-60s %s  best %.4f ms" % (k, " ".join("%.4f" % x for x in t),
                                          min(t)))


if __name__ == "__main__":
    main()
