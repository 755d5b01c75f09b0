"""Is the trace kernel power limited, and does it matter?  Session 4 found the
socket at 1375 W of its 1400 W limit with the power limiter active ~60 % of
the time while the headline kernel runs.  Here every kind of launch runs
back to back for a few seconds (laboratory build: probes) and the steady
state is recorded: mean launch time, gfx clock, socket power, share of the
time the power limiter (PPT) was active."""
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
S = len(system) - 1


def steady(name, launch, bytes_per_launch, seconds=4.):
    tele = Telemetry(0, period=0.01)
    t_end = time.time() + seconds
    ms = []
    k = 0
    while time.time() < t_end:
        v = launch()
        if v is not None:
            ms.append(v)
        k += 1
        if k == 40:     # the first second is the ramp of the power filter
            tele.mark("steady:begin")
    tele.mark("steady:end")
    time.sleep(.05)
    t = tele.stop()
    w = (t or {}).get("steady") or {}
    m = float(np.median(ms[len(ms)//3:]))
    print(json.dumps({
        "what": name, "launch_ms": m,
        "TBs": bytes_per_launch/m/1e9 if bytes_per_launch else None,
        "gfxclk_mhz": (w.get("gfxclk_mhz") or [None]*3)[1],
        "socket_power_w": (w.get("socket_power_w") or [None]*3)[1],
        "power_limited_fraction": w.get("power_limited_fraction"),
        "hotspot_c": (w.get("hotspot_c") or [None]*3)[2],
        "hbm_c": (w.get("hbm_c") or [None]*3)[2]}), flush=True)
    time.sleep(3.)


def traces(k=10, **kw):
    def launch():
        eng.event_record(0)
        for _ in range(k):
            eng.trace(1, 0, True)
        eng.event_record(1)
        return eng.event_elapsed(0, 1)/k
    return launch


def probe(mode):
    def launch():
        return eng.probe(mode)[0]
    return launch


steady("trace kernel, C3 host-seeded", traces(), n*(56*S + 48))
steady("store pattern + input read, no arithmetic (probe 7)", probe(7),
       n*(56*S + 48))
steady("store pattern, no read, no arithmetic (probe 8)", probe(8), n*56*S)
steady("fill, one 16-byte store per lane (probe 3)", probe(3), None)
steady("16-byte copy (probe 2)", probe(2), None)
g.propagate(clip=True, keep=[0, -1])
steady("trace kernel, image row only (FP64 side alone)", traces(), None)
g.propagate(clip=True)
for pad in (20480, 40960, 65536):
    eng.set_option("lds_pad", pad)
    steady("trace kernel, lds_pad=%d (fewer resident wavefronts)" % pad,
           traces(), n*(56*S + 48))
eng.set_option("lds_pad", 0)
# C4: exact against the default arithmetic
import digest_cases as dc
s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
y4, u4 = dc.bundle(n, .6, 10., 4)
y4[:, 1] -= .5*np.tan(np.radians(10.))
S4 = len(s4) - 1
for label, opts in (("default arithmetic", {}), ("exact_asphere", {"exact_asphere": 1})):
    g4 = ra.GeometricTrace(s4, **opts)
    g4.rays_given(y4, u4)
    g4.propagate(clip=True)
    e4 = g4.engine

    def launch():
        e4.event_record(0)
        for _ in range(10):
            e4.trace(1, 0, True)
        e4.event_record(1)
        return e4.event_elapsed(0, 1)/10
    steady("C4 asphere, " + label, launch, n*(56*S4 + 48))
    del g4
