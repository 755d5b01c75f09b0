"""How close is the device to the oracle (== the reference)?  Prints, per
BASELINE config, the largest deviation of y,u,i,t in units of the largest
finite magnitude of the same surface row, and the number of NaN-mask
mismatches.  The contract is 1e-10 (sphere/conic) / 1e-8 (aspheres)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.bundles import disc_bundle, multi_field_bundle
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn


def report(name, system, y, u, l=None, clip=True):
    g = ra.GeometricTrace(system)
    g.rays_given(y, u, l)
    g.propagate(clip=clip)
    table, ns = pack_system(system, g.l, g.n[0])
    with np.errstate(all="ignore"):
        want = tn.propagate(table, y, u, clip=clip)
    worst, mism, exact = 0., 0, 0
    total = 0
    for rows, b in zip((g.y, g.u, g.i, g.t), want):
        a = np.asarray(rows[1:])
        mism += int((np.isnan(a) != np.isnan(b)).sum())
        for j in range(b.shape[0]):
            fin = np.isfinite(b[j]) & np.isfinite(a[j])
            if not fin.any():
                continue
            scale = np.abs(b[j][fin]).max()
            worst = max(worst, float(np.abs(a[j][fin] - b[j][fin]).max()/scale))
            exact += int((a[j][fin] == b[j][fin]).sum())
            total += int(fin.sum())
    print("%-34s rays %8d  max dev %.2e  bit-identical values %.4f %%  "
          "NaN-mask mismatches %d" % (name, len(y), worst, 100.*exact/total,
                                      mism), flush=True)


def main():
    report("C1 singlet", ra.system_from_yaml(P.SINGLET),
           *disc_bundle(10**4, 8., 0., 0))
    for l in (587.56e-9, 656.27e-9, 486.13e-9):
        report("C2 cooke %.0f nm" % (l*1e9), ra.system_from_yaml(P.cooke(l)),
               *disc_bundle(10**6, 5.5, 5., 0), l=l)
    th = [f*P.DOUBLE_GAUSS_FIELD_DEG for f in (0, .35, .5, .7, 1.)]
    report("C3 double-gauss", ra.system_from_yaml(P.DOUBLE_GAUSS),
           *multi_field_bundle(2*10**6, 17., th, 0, P.DOUBLE_GAUSS_PUPIL_Z))
    y, u = disc_bundle(3*10**5, 0.6, 17.5, 3)
    y[:, 1] -= 0.5*np.tan(np.radians(17.5))
    report("C4 asphere phone lens", ra.system_from_yaml(P.ASPHERE_PHONE), y, u)
    report("torture (tilts, conics, mirror)", ra.system_from_yaml(P.TORTURE),
           *disc_bundle(10**6, 12., 3., 1))


if __name__ == "__main__":
    main()
