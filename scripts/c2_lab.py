#!/usr/bin/env python
"""C2's lottery (round 6): contexts of the Cooke batch (3 x 10^6 rays, 2.2 GB)
per piece size (RT_MI355_PIECE_MIB; a process each, RT_MI355_PLACE_LOG on):
what every search saw -- classes in creation order, pair ratios, the store
pattern over each set -- and the settled launch time.

    python scripts/c2_lab.py contexts mib [mib ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def child(reps):
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    import digest_cases as dc
    import bench_legs as legs
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    y, u = dc.bundle(10**6, 5.5, 5., 0)
    if os.environ.get("LAB_SHAPE") == "C3":
        s2 = ra.system_from_yaml(P.DOUBLE_GAUSS)
        ls = None
        y, u = legs.workload_rays(10_000_000, 7)
    for k in range(reps):
        g = ra.GeometricTrace(s2)
        g.rays_given(y, u, ls)
        ms = legs.kernel_ms_of(g, True, settle_s=.2, dwell_s=.25)
        pl = g.engine.placement()
        print(json.dumps({
            "piece_mib": pl["piece_mib"], "pieces": pl["pieces"],
            "per_class": pl["per_class"], "trace_ms": round(ms, 4),
            "sets": [round(v) for v in
                     pl["store_pattern_GBps_per_piece_set"]],
            "search_ms": round(pl["search_ms"]["reserve_total"], 1)}),
            flush=True)
        del g


if __name__ == "__main__":
    if sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        reps = int(sys.argv[1])
        for mib in sys.argv[2:]:
            env = dict(os.environ, RT_MI355_PLACE_LOG="1")
            if int(mib):
                env["RT_MI355_PIECE_MIB"] = mib
            print("== piece MiB", mib, flush=True)
            subprocess.run([sys.executable, __file__, "--child", str(reps)],
                           env=env, stderr=subprocess.STDOUT)
