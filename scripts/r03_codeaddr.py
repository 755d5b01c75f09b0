"""Does the launch time depend on WHERE the code object is loaded?  The same
library file copied under eight names and loaded eight times in one process
(eight code objects at eight addresses), one context each, same rays: steady
launch time per instance, two rounds."""
import json, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
tmp = tempfile.mkdtemp(prefix="rt_codeaddr_")


def steady(eng, seconds=1.2):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


traces = []
for k in range(8):
    path = os.path.join(tmp, "librt_copy%d.so" % k)
    shutil.copy(_build.LIB, path)
    g = ra.GeometricTrace(system, engine=Engine(0, lib_path=path))
    g.rays_given(y, u)
    g.propagate(clip=True)
    traces.append(g)
steady(traces[0].engine, 2.)
res = [[] for _ in traces]
for rep in range(2):
    order = list(range(len(traces)))
    for k in (order if rep == 0 else order[::-1]):
        res[k].append(steady(traces[k].engine))
for k in range(len(traces)):
    print(json.dumps({"instance": k, "steady_ms": res[k]}), flush=True)
v = [np.mean(r) for r in res]
print(json.dumps({"min_ms": float(min(v)), "max_ms": float(max(v)),
                  "spread": float(max(v)/min(v) - 1.)}))
shutil.rmtree(tmp, ignore_errors=True)
