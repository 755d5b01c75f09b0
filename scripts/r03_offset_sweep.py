"""The launch time at four workgroups per CU depends on which allocation the
arrays live in (r03_placement.py).  Is it the start ADDRESS?  One allocation
(laboratory build: 1 GiB larger than the arrays), the arrays moved through it
by the option base_offset_kb; launch time at two and at four workgroups per CU
for each offset.  Offsets: fine steps (4 KiB .. 2 MiB) and coarse ones
(2 MiB .. 1 GiB)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from rayopt_amd._lib import RT_Y
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
lab = os.path.join(os.path.dirname(_build.LIB), "librt_mi355_probes.so")


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


offsets = [0, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 1536,
           2048, 4096, 6144, 8192, 16384, 32768, 65536, 131072, 262144,
           524288, 786432, 1048576 - 2048, 0]
for inst in range(2):           # two allocations in this process
    g = ra.GeometricTrace(system, engine=Engine(0, lib_path=lab))
    eng = g.engine
    g.rays_given(y, u)
    g.propagate(clip=True)
    steady(eng, 2. if inst == 0 else .5)
    for kb in offsets:
        eng.set_option("base_offset_kb", kb)
        g.rays_given(y, u)
        g.propagate(clip=True)
        res = {}
        for lds in (65536, 32768):
            eng.set_option("resident_lds", lds)
            res[str(lds)] = steady(eng, .45)
        eng.set_option("resident_lds", -1)
        print(json.dumps({"allocation": inst, "base_offset_kb": kb,
                          "Y_row1_address": hex(eng.device_ptr(RT_Y, 1)),
                          "steady_ms_by_resident_lds": res}), flush=True)
