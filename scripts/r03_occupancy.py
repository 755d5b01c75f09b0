"""Session 5 found the trace kernel 5 % FASTER with two workgroups per CU
(unused dynamic LDS) than with seven.  Sweep: resident workgroups per CU
(lds_pad) x workgroup size, steady state, C3 host-seeded; laboratory build."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays, Telemetry

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(system)
g.rays_given(y, u)
eng = g.engine
g.propagate(clip=True)
S = len(system) - 1
B = n*(56*S + 48)


TELE = {}


def steady(seconds=1.5):
    tele = Telemetry(0, period=0.01)
    t_end = time.time() + seconds
    ms = []
    k = 0
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
        k += 1
        if k == 30:
            tele.mark("steady:begin")
    tele.mark("steady:end")
    time.sleep(.03)
    w = ((tele.stop() or {}).get("steady")) or {}
    TELE.clear()
    TELE.update(gfxclk_mhz=(w.get("gfxclk_mhz") or [None]*3)[1],
                socket_power_w=(w.get("socket_power_w") or [None]*3)[1],
                power_limited_fraction=w.get("power_limited_fraction"),
                hotspot_c=(w.get("hotspot_c") or [None]*3)[2])
    return float(np.median(ms[len(ms)//3:]))


# warm the chip into its power-limited steady state first
steady(4.)
for rep in range(2):
    for block, pads in ((256, (0, 32768, 40960, 53248, 65536)),
                        (128, (0, 24576, 32768, 40960))):
        for pad in pads:
            eng.set_option("block", block)
            eng.set_option("lds_pad", pad)
            ms = steady()
            per_cu = 7 if not pad else min(7, 163840//pad)
            print(json.dumps({"rep": rep, "block": block, "lds_pad": pad,
                              "workgroups_per_cu_by_lds": per_cu,
                              "waves_per_cu": per_cu*block//64,
                              "launch_ms": ms, "TBs": B/ms/1e9, **TELE}),
                  flush=True)
eng.set_option("block", 256)
eng.set_option("lds_pad", 0)
print(json.dumps({"baseline_again_ms": steady()}))
