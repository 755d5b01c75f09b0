"""Workgroups resident per CU (engine option "resident_lds") per kind of trace:
where does capping the trace kernel at two workgroups per CU pay, where does
it cost?  Shipped library, steady state, telemetry per cell."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import (workload_rays, Telemetry, FIELD_FRACTIONS, BUNDLE_RADIUS)
import digest_cases as dc

n = 10_000_000


def steady(g, clip, keep=None, seconds=1.3):
    tele = Telemetry(0, period=0.01)
    eng = g.engine
    t_end = time.time() + seconds
    ms, k = [], 0
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, clip)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
        k += 1
        if k == 25:
            tele.mark("steady:begin")
    tele.mark("steady:end")
    time.sleep(.03)
    w = ((tele.stop() or {}).get("steady")) or {}
    return float(np.median(ms[len(ms)//3:])), dict(
        gfxclk_mhz=(w.get("gfxclk_mhz") or [None]*3)[1],
        socket_power_w=(w.get("socket_power_w") or [None]*3)[1],
        power_limited_fraction=w.get("power_limited_fraction"))


def sweep(name, g, clip, keep=None, pads=(0, 65536, 40960, 32768)):
    g.propagate(clip=clip, keep=keep)
    for rep in range(2):
        for pad in pads:
            g.engine.set_option("resident_lds", pad)
            ms, t = steady(g, clip)
            print(json.dumps(dict(what=name, rep=rep, resident_lds=pad,
                                  launch_ms=ms, **t)), flush=True)
    g.engine.set_option("resident_lds", -1)


s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
g = ra.GeometricTrace(s3)
g.rays_given(y, u)
steady(g, True, seconds=3.)        # into the power-limited steady state
sweep("C3 host-seeded, all rows", g, True)
sweep("C3 host-seeded, unclipped", g, False, pads=(0, 65536))
sweep("C3 image row only (FP64 side)", g, True, keep=[0, -1],
      pads=(0, 65536, 32768))
del g
nf = len(FIELD_FRACTIONS)
m = n//nf//64*64
g = ra.GeometricTrace(s3)
g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS], dc.disc_points(m, 7),
              P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
sweep("C3 device-generated (regen kernel)", g, True)
del g
s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
y4, u4 = dc.bundle(n, .6, 10., 4)
y4[:, 1] -= .5*np.tan(np.radians(10.))
for label, opts in (("default arithmetic", {}),
                    ("exact_asphere", {"exact_asphere": 1})):
    g = ra.GeometricTrace(s4, **opts)
    g.rays_given(y4, u4)
    sweep("C4 asphere, " + label, g, True)
    del g
st = ra.system_from_yaml(P.TORTURE)
yt, ut = dc.bundle(n, 12., 2., 3)
g = ra.GeometricTrace(st)
g.rays_given(yt, ut)
sweep("torture (tilted: i rows stored, 80 B/op)", g, True, pads=(0, 65536))
del g
s2 = ra.system_from_yaml(P.COOKE % dict(air="air", sk16="SCHOTT-SK|N-SK16",
                                        f2="SCHOTT-F|N-F2"))
y2, u2 = dc.bundle(10**6, 5.5, 5., 0)
g = ra.GeometricTrace(s2)
g.rays_given(y2, u2, l=[587.56e-9, 656.27e-9, 486.13e-9])
sweep("C2 Cooke 3 x 10^6 rays, one launch", g, True, pads=(0, 65536))
