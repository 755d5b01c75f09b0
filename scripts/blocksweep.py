"""Why one generated batch of 10^8 rays takes 3-5 % longer per ray than
batches of 10^7 (round 5): the double-Gauss batch of
bench.py's C5 leg -- N rays built on the device as F field bundles over the
SAME N/F pupil points -- traced with the automatic block plan or with blocks
of a given number of rays (option "block_rays"), each case in a context of
its own: settled launch time, per 10^7 rays, with what the placement measured.
The size of the pupil-point array (16 B per point, read once per bundle) is
what F varies at a fixed N.

    python scripts/blocksweep.py N:F[:block_rays[:turn_points]] ...

turn_points (rt_set_option): 0 automatic, -1 the ray order, P turns of P
points (csrc/rt_lay.h: rt_gen_wg).

    python scripts/blocksweep.py --ab N:F turn_points ...

ONE context (one set of pieces), the option switched between measurements.
"""
import gc
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

import rayopt_amd as ra                                    # noqa: E402
from rayopt_amd import prescriptions as P                  # noqa: E402
import digest_cases as dc                                  # noqa: E402
from bench_legs import FIELD_FRACTIONS, BUNDLE_RADIUS      # noqa: E402
from boxstat import settled_ms                             # noqa: E402


def ab(case, turns):
    s = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n, nf = (int(float(v)) for v in case.split(":"))
    m = n//nf//64*64
    g = ra.GeometricTrace(s)
    g.rays_fields(np.c_[np.zeros(nf),
                        np.linspace(0., max(FIELD_FRACTIONS), nf)],
                  dc.disc_points(m, 91), P.DOUBLE_GAUSS_PUPIL_Z,
                  BUNDLE_RADIUS)
    pl = None
    for turn in turns:
        g.engine.set_option("turn_points", turn)
        ms = settled_ms(g, settle_s=.3, dwell_s=.5)
        pl = pl or g.engine.placement()
        print(json.dumps({
            "rays": m*nf, "bundles": nf, "pupil_points": m,
            "turn_points": turn, "trace_ms": ms,
            "ms_per_1e7_rays": ms*1e7/(m*nf), "same_context": True,
            "per_class": pl["per_class"],
            "store_pattern_GBps": pl["store_pattern_GBps"]}), flush=True)


def main():
    if sys.argv[1:2] == ["--ab"]:
        return ab(sys.argv[2], [int(float(v)) for v in sys.argv[3:]])
    s = ra.system_from_yaml(P.DOUBLE_GAUSS)
    top = max(FIELD_FRACTIONS)
    for case in sys.argv[1:] or ["1e8:5", "1e8:50", "1e7:5", "1e7:1"]:
        part = case.split(":")
        n, nf = int(float(part[0])), int(part[1])
        b = int(float(part[2])) if len(part) > 2 else 0
        turn = int(float(part[3])) if len(part) > 3 else 0
        m = n//nf//64*64
        pts = dc.disc_points(m, 91)
        fields = np.c_[np.zeros(nf), np.linspace(0., top, nf)]
        g = ra.GeometricTrace(s, turn_points=turn,
                              **({"block_rays": b} if b else {}))
        g.rays_fields(fields, pts, P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
        ms = settled_ms(g, settle_s=.5, dwell_s=.6)
        pl = g.engine.placement()
        print(json.dumps({
            "rays": m*nf, "bundles": nf, "pupil_points": m,
            "pupil_point_bytes": 16*m, "block_rays_asked": b, "turn_points": turn,
            "blocks": g.engine.blocks(), "trace_ms": ms,
            "ms_per_1e7_rays": ms*1e7/(m*nf), "pieces": pl["pieces"],
            "per_class": pl["per_class"],
            "store_pattern_GBps": pl["store_pattern_GBps"]}), flush=True)
        del g
        gc.collect()


if __name__ == "__main__":
    main()
