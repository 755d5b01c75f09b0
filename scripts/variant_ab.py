#!/usr/bin/env python
"""Launch times of the three trace shapes -- host-seeded headline batch, the
same bundles built on the device, C4 on the default arithmetic -- per build of
the library (RT_MI355_LIB; a process each), in turn, `rounds` times.

    python scripts/variant_ab.py rounds lib1.so lib2.so ...
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def child():
    import numpy as np
    import rayopt_amd as ra
    from rayopt_amd import prescriptions as P
    import bench_legs as legs
    import digest_cases as dc
    out = {"lib": os.path.basename(os.environ.get("RT_MI355_LIB", "product"))}
    system = ra.system_from_yaml(P.DOUBLE_GAUSS)
    n = 10_000_000
    y, u = legs.workload_rays(n, 0)
    g = ra.GeometricTrace(system)
    g.rays_given(y, u)
    out["headline_ms"] = round(legs.kernel_ms_of(g, True), 4)
    out["headline_GBps"] = round(g.engine.placement()["store_pattern_GBps"])
    del g
    nf = len(legs.FIELD_FRACTIONS)
    h = ra.GeometricTrace(system)
    h.rays_fields(np.c_[np.zeros(nf), legs.FIELD_FRACTIONS],
                  dc.disc_points(n//nf//64*64, 77), P.DOUBLE_GAUSS_PUPIL_Z,
                  legs.BUNDLE_RADIUS)
    h.propagate(clip=True)
    out["generated_ms"] = round(legs.kernel_ms_of(h, True), 4)
    out["generated_GBps"] = round(h.engine.placement()["store_pattern_GBps"])
    del h
    s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
    y, u = dc.bundle(n, .6, 10., 4)
    y[:, 1] -= .5*np.tan(np.radians(10.))
    for key, opts in (("c4_ms", {}), ("c4x_ms", {"exact_asphere": 1})):
        k = ra.GeometricTrace(s4, **opts)
        k.rays_given(y, u, s4.wavelengths[0])
        out[key] = round(legs.kernel_ms_of(k, True), 4)
        del k
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        rounds = int(sys.argv[1])
        libs = sys.argv[2:]
        for r in range(rounds):
            for lib in libs:
                env = dict(os.environ)
                if lib != "product":
                    env["RT_MI355_LIB"] = os.path.abspath(lib)
                res = subprocess.run([sys.executable, __file__, "--child"],
                                     env=env, capture_output=True, text=True)
                sys.stdout.write(res.stdout or json.dumps(
                    {"lib": lib, "error": res.stderr[-400:]}) + "\n")
                sys.stdout.flush()
