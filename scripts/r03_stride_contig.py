"""Row stride sweep on a physically CONTIGUOUS allocation (laboratory build,
alloc_round = 99: a power-of-two allocation is one buddy block).  On such an
allocation four workgroups per CU were slow in 16 of 16 cases
(r03_alloc_round.py), so the launch time should be a function of the row
stride alone: n = 10^7 + 64 j rays, i.e. the stride grows in steps of 512 B."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from bench import workload_rays

LAB = os.path.join(os.path.dirname(_build.LIB), "librt_mi355_probes.so")
n0 = 10_000_000
pads = list(range(0, 64)) + [64, 96, 128, 192, 256, 384, 512, 768, 1024,
                             2048, 4096, 8192, 16384, 32768]
nmax = n0 + 64*max(pads)
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(nmax - nmax % 5, 0)
y = np.concatenate([y, y[:5]])[:nmax]
u = np.concatenate([u, u[:5]])[:nmax]
eng = Engine(0, lib_path=LAB)
eng.set_option("alloc_round", 99)
g = ra.GeometricTrace(system, engine=eng)
g.rays_given(y, u)              # the largest first: one allocation for all
g.propagate(clip=True)


def steady(seconds):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


steady(2.)
for j in pads + [0]:
    n = n0 + 64*j
    g.rays_given(y[:n], u[:n])
    g.propagate(clip=True)
    res = {}
    for lds in (65536, 32768):
        eng.set_option("resident_lds", lds)
        res[str(lds)] = steady(.4)*n0/n     # per 10^7 rays
    eng.set_option("resident_lds", -1)
    print(json.dumps({"pad_rays": 64*j, "stride_bytes": 8*eng.ld,
                      "stride_mod_4096": 8*eng.ld % 4096,
                      "ms_per_1e7_rays_by_resident_lds": res}), flush=True)
