"""Arrays in virtual memory backed by 1 GiB physical chunks (laboratory build,
alloc_vmm_mb): the SAME chunks behind the SAME addresses in different orders
(alloc_vmm_seed) -- does the order decide the launch time?  argv: chunk MiB,
number of contexts, number of orders."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from bench import workload_rays

LAB = os.path.join(os.path.dirname(_build.LIB), "librt_mi355_probes.so")
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nctx = int(sys.argv[2]) if len(sys.argv) > 2 else 2
nseed = int(sys.argv[3]) if len(sys.argv) > 3 else 8
n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)


def steady(eng, seconds):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


keep = []
for c in range(nctx):
    eng = Engine(0, lib_path=LAB)
    eng.set_option("alloc_vmm_mb", mb)
    eng.set_option("alloc_vmm_shuffle", 1)
    g = ra.GeometricTrace(system, engine=eng)
    keep.append(g)
    g.rays_given(y, u)
    g.propagate(clip=True)
    if c == 0:
        steady(eng, 2.)
    for seed in list(range(nseed)) + [0]:
        eng.set_option("alloc_vmm_seed", seed)
        g.rays_given(y, u)
        g.propagate(clip=True)
        res = {}
        for lds in (65536, 32768, 0):
            eng.set_option("resident_lds", lds)
            res[str(lds)] = steady(eng, .4)
        eng.set_option("resident_lds", -1)
        print(json.dumps({"chunk_mib": mb, "context": c, "order": seed,
                          "steady_ms_by_resident_lds": res}), flush=True)
