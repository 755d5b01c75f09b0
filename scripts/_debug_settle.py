import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T
from bench import workload_rays

system = ra.system_from_yaml(P.DOUBLE_GAUSS)
L = len(system)
n = 3_000_000
y, u = workload_rays(n, 0)
rows = {}
for good in (0, 10**6):
    eng = ra.Engine()
    eng.set_option("placement_good_gbps", good)
    eng.set_option("block_rays", 1_600_000)
    g = ra.GeometricTrace(system, engine=eng)
    g.rays_given(y, u)
    g.propagate(clip=True)
    print(good, "pass 1", eng.placement()["store_pattern_GBps_per_piece_set"], eng.blocks(), flush=True)
    g.rays_given(y[:2_000_000], u[:2_000_000])
    print(good, "after 2nd rays_given", eng.placement()["store_pattern_GBps_per_piece_set"], eng.placement()["store_pattern_GBps_per_range"], eng.blocks(), flush=True)
    r0 = [eng.download(w, 0, 1) for w in (RT_Y, RT_U)]
    g.propagate(clip=True)
    rows[good] = [eng.download(w, 0, L) for w in (RT_Y, RT_U, RT_I, RT_T)]
    rows[good, "r0"] = r0
    # trace again: does a second propagate repair it?
    g.propagate(clip=True)
    rows[good, "again"] = [eng.download(w, 0, L) for w in (RT_Y, RT_U, RT_I, RT_T)]
    eng.close()
for k, name in enumerate("YU"):
    a, b = rows[0, "r0"][k], rows[10**6, "r0"][k]
    print("row 0 of", name, "equal:", np.array_equal(a, b, equal_nan=True))
for tag in (None, "again"):
    A = rows[0] if tag is None else rows[0, tag]
    B = rows[10**6] if tag is None else rows[10**6, tag]
    for name, a, b in zip("YUIT", A, B):
        bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
        if bad.any():
            idx = np.argwhere(bad)
            print(tag, name, "differ:", bad.sum(), "of", bad.size, "rows", sorted(set(idx[:, 0]))[:14],
                  "rays", idx[:, -1].min(), "..", idx[:, -1].max(), "sample", a[tuple(idx[0])], b[tuple(idx[0])])
        else:
            print(tag, name, "equal")
