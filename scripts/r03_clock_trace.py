"""What moves the launch time between "plateaus"?  The headline kernel (C3,
10^7 rays, host-seeded) launched back to back for ~45 s while a child process
samples amdsmi (every metric it offers) every 10 ms; per block of 10 launches
the mean launch time from HIP events.  Then the same with the device left idle
for 5 s in between, to see whether the slow state is reached by load or by
time."""
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
eng.sync()

try:
    import amdsmi
    amdsmi.amdsmi_init()
    h = amdsmi.amdsmi_get_processor_handles()[0]
    m = amdsmi.amdsmi_get_gpu_metrics_info(h)
    print(json.dumps({"all_metric_keys": {k: (v if not isinstance(v, list)
                                              else v[:8]) for k, v in m.items()}},
                     default=str), flush=True)
    amdsmi.amdsmi_shut_down()
except Exception as err:
    print(json.dumps({"amdsmi": repr(err)}), flush=True)

tele = Telemetry(0, period=0.01)
t_origin = time.time()
series = []


def burst(seconds, label):
    t_end = time.time() + seconds
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms = eng.event_elapsed(0, 1)/10
        series.append((time.time() - t_origin, ms, label))


tele.mark("run:begin")
burst(25., "continuous")
time.sleep(5.)
burst(8., "after 5 s idle")
time.sleep(20.)
burst(8., "after 20 s idle")
tele.mark("run:end")
t = tele.stop(raw=True)
t['t_origin'] = t_origin
print(json.dumps({"launch_series": series}), flush=True)
print(json.dumps({"telemetry_summary": t}), flush=True)
