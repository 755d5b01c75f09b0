"""rayopt itself against the numpy port (oracle/trace_numpy.py) on THIS host,
same rays, one core each: the evidence behind bench.py's cpu_baseline
``kind: "port"`` on boxes that do not have /root/reference (the GPU pool).
Needs no GPU.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.pack import pack_system
from bench import workload_rays
from oracle import trace_numpy as tn, ref_timing


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    l = system.wavelengths[0]
    y, u = workload_rays(n, 0)
    table, _ = pack_system(system, l, system.refractive_index(l, 0))
    S = len(system) - 1
    tn.propagate(table, y[:100000], u[:100000], clip=True)
    t0 = time.perf_counter()
    Y, U, I, T = tn.propagate(table, y, u, clip=True)
    dt = time.perf_counter() - t0
    port = {"value": n*S/dt, "unit": "ray-surface-ops/s", "cores": 1,
            "kind": "port", "seconds": dt, "rays": n}
    ref = ref_timing.time_reference(y, u, l, True, want_image_row=Y[-1],
                                    max_rays=n)
    out = {"host_cores": os.cpu_count(), "port": port, "reference": ref,
           "port_over_reference": (port["value"]/ref["value"]
                                   if ref else None)}
    # the asphere config -- the reference solves every ray
    # with its own scipy.optimize.newton call (rayopt/elements.py:333-349)
    asph = ra.system_from_yaml(P.ASPHERE_PHONE)
    la = asph.wavelengths[0]
    ta, _ = pack_system(asph, la, asph.refractive_index(la, 0))
    Sa = len(asph) - 1
    m = 200_000
    ya, ua = ra.bundles.disc_bundle(m, 0.6, 17.5, 3)
    ya[:, 1] -= 0.5*np.tan(np.radians(17.5))
    tn.propagate(ta, ya[:1000], ua[:1000], clip=True)
    t0 = time.perf_counter()
    Ya = tn.propagate(ta, ya, ua, clip=True)[0]
    dt = time.perf_counter() - t0
    out["asphere_port"] = {"value": m*Sa/dt, "rays": m, "seconds": dt,
                           "note": "C4, numpy port (masked vector Newton)"}
    k = 3000
    refa = ref_timing.time_reference(ya[:k], ua[:k], la, True,
                                     want_image_row=None, max_rays=k,
                                     prescription=P.ASPHERE_PHONE)
    if refa:
        refa["note"] = "C4, rayopt itself: one scipy newton call per ray"
    out["asphere_reference"] = refa
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
