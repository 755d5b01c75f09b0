"""Kernel time of every BASELINE.json config on one GPU (documentation table;
the headline bench line is bench.py).  Prints one JSON object per config."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd._lib import F_ROTATED, F_REFRACT
from rayopt_amd.bundles import disc_bundle, multi_field_bundle
from rayopt_amd.pack import pack_system


def run(name, system, y, u, l, clip, reps=60, keep=None, **options):
    g = ra.GeometricTrace(system, **options)
    g.rays_given(y, u, l)
    ms = []
    for k in range(reps):       # the first ~50 launches ride the clock ramp
        g.propagate(clip=clip, keep=keep)
        ms.append(g.kernel_ms())
    ms = float(np.median(ms[-10:]))
    n, S = y.shape[0], len(system) - 1
    table, _ = pack_system(system, g.l, g.n[0])
    rot = (table["flags"] & F_ROTATED) != 0
    bends = (table["flags"] & F_REFRACT) != 0
    stored_i = sum(1 for j in range(1, S + 1) if rot[j] or rot[j - 1])
    skipped_u = 0 if clip else sum(1 for j in range(1, S + 1)
                                   if not bends[j])
    nbytes = n*(56*S + 24*stored_i - 24*skipped_u + 48)
    dead = float(np.isnan(np.asarray(g.u[-1])[:, 0]).mean())
    rec = dict(config=name, rays=n, surfaces=S, clip=clip, kernel_ms=ms,
               ops_per_s=n*S/ms*1e3, dead_fraction=dead)
    if keep is None:
        rec["GBs"] = nbytes/ms/1e6
    rec.update(options)
    print(json.dumps(rec), flush=True)
    g.engine.close()


def main():
    s = ra.system_from_yaml(P.SINGLET)
    run("C1 singlet 1e4", s, *disc_bundle(10**4, 8., 0., 0), None, True)
    for l in (587.56e-9, 656.27e-9, 486.13e-9):
        s = ra.system_from_yaml(P.cooke(l))
        run("C2 cooke 1e6 l=%.0fnm" % (l*1e9), s,
            *disc_bundle(10**6, 5.5, 5., 0), l, True)
    # C2 as ONE launch: the same 10^6 rays at the three wavelengths (ray
    # groups, one surface table each) through the dispersive prescription
    s = ra.system_from_yaml(P.COOKE % dict(air=1.0, sk16="1.62041/60.32",
                                           f2="1.62004/36.37"))
    y, u = disc_bundle(10**6 - 10**6 % 64, 5.5, 5., 0)
    g = ra.GeometricTrace(s)
    g.rays_given(y, u, l=[587.56e-9, 656.27e-9, 486.13e-9])
    ms = []
    for k in range(60):
        g.propagate(clip=True)
        ms.append(g.kernel_ms())
    ms = float(np.median(ms[-10:]))
    n3, S = 3*len(y), len(s) - 1
    print(json.dumps(dict(config="C2 cooke, 3 wavelengths x 1e6 rays in ONE "
                          "launch", rays=n3, surfaces=S, clip=True,
                          kernel_ms=ms, ops_per_s=n3*S/ms*1e3,
                          GBs=n3*(56*S + 48)/ms/1e6)), flush=True)
    g.engine.close()
    s = ra.system_from_yaml(P.DOUBLE_GAUSS)
    th = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    y, u = multi_field_bundle(10**7, 17., th, 0, P.DOUBLE_GAUSS_PUPIL_Z)
    run("C3 double-gauss 1e7 clip", s, y, u, None, True)
    run("C3 double-gauss 1e7 noclip", s, y, u, None, False)
    s = ra.system_from_yaml(P.ASPHERE_PHONE)
    for deg in (0., 17.5):
        y, u = disc_bundle(10**7, 0.6, deg, 3)
        y[:, 1] -= 0.5*np.tan(np.radians(deg))
        run("C4 asphere 1e7 field %.1f deg, exact Newton (bit-identical to "
            "the reference)" % deg, s, y, u, None, True)
        run("C4 asphere 1e7 field %.1f deg, fast_asphere (1e-8 contract)"
            % deg, s, y, u, None, True, fast_asphere=1)
    # over-filled C3 (bundle radius x1.5: 55 % of the rays vignette): every
    # row stored, image row only, image row only with the compacting kernel
    s = ra.system_from_yaml(P.DOUBLE_GAUSS)
    y, u = multi_field_bundle(10**7, 17.*1.5, th, 0, P.DOUBLE_GAUSS_PUPIL_Z)
    run("C3 overfilled x1.5, all rows", s, y, u, None, True)
    run("C3 overfilled x1.5, image row only", s, y, u, None, True,
        keep=[0, -1])
    run("C3 overfilled x1.5, image row only, compacting kernel", s, y, u,
        None, True, keep=[0, -1], compact=1)
    s = ra.system_from_yaml(P.TORTURE)
    run("torture (tilts, conics, mirror) 1e7", s, *disc_bundle(10**7, 9., 2., 1),
        None, True)
    # C5's whole batch on ONE GPU: 10^8 rays x 13 elements = 104 GB of result
    # arrays in HBM; rays built on the device (5 fields x 2*10^7 pupil points)
    s = ra.system_from_yaml(P.DOUBLE_GAUSS)
    rng = np.random.default_rng(0)
    m = 2*10**7
    r, phi = np.sqrt(rng.random(m)), 2*np.pi*rng.random(m)
    yp = np.c_[r*np.cos(phi), r*np.sin(phi)]
    fields = np.c_[np.zeros(5), [0, .35, .5, .7, 1.]]
    g = ra.GeometricTrace(s)
    g.rays_fields(fields, yp, P.DOUBLE_GAUSS_PUPIL_Z, 17.)
    gen_ms = g.kernel_ms()
    ms = []
    for k in range(12):
        g.propagate(clip=True)
        ms.append(g.kernel_ms())
    ms = float(np.median(ms[-5:]))
    n, S = g.nrays, len(s) - 1
    print(json.dumps(dict(config="C5 batch on one GPU: 1e8 rays, device-"
                          "generated", rays=n, surfaces=S, clip=True,
                          kernel_ms=ms, ops_per_s=n*S/ms*1e3,
                          GBs=n*(56*S + 56)/ms/1e6, generate_ms=gen_ms,
                          result_GB=n*13*80/1e9)), flush=True)


if __name__ == "__main__":
    main()
