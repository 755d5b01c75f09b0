"""Does the row stride matter?  The result rows of the SoA layout are ld*8
bytes apart (ld = rays rounded up to 64): 84 row streams written side by
side, 7 of them by every store burst of a wavefront.  If HBM channels / banks
are selected from address bits that the stride leaves equal, the streams of
one burst collide.  Sweep the stride by tracing n = 10^7 + pad rays: the real
kernel (host-seeded C3) and its store pattern without arithmetic (rt_probe
modes 7 = with the input read, 8 = without)."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from bench import workload_rays

base = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(base + 70000, 0)
g = ra.GeometricTrace(system)
pads = [0, 64, 128, 192, 256, 320, 512, 1024, 1536, 2048, 4096, 4160, 8192,
        16384, 32768, 65536, 5056, 12352]
S = len(system) - 1
for rep in range(2):
    for pad in pads:
        n = base + pad
        g.rays_given(y[:n], u[:n])
        eng = g.engine
        for _ in range(25 if rep == 0 and pad == 0 else 6):
            g.propagate(clip=True)
        t = []
        for _ in range(12):
            g.propagate(clip=True)
            t.append(g.kernel_ms())
        ms = float(np.median(t))
        out = dict(rep=rep, pad=pad, ld=eng.ld, stride_mod_4096=eng.ld*8 % 4096,
                   kernel_ms=ms, TBs=n*(56*S + 48)/ms/1e9)
        for mode in (7, 8):
            tt = []
            for _ in range(8):
                pm, b = eng.probe(mode)
                tt.append(pm)
            pm = float(np.median(tt[2:]))
            out["probe%d_ms" % mode] = pm
            out["probe%d_TBs" % mode] = b/pm/1e9
        print(json.dumps(out), flush=True)
