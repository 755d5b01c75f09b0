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
    print(json.dumps({"host_cores": os.cpu_count(), "port": port,
                      "reference": ref,
                      "port_over_reference": (port["value"]/ref["value"]
                                              if ref else None)}, indent=1))


if __name__ == "__main__":
    main()
