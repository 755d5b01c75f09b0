"""Same-process A/B: the element's flag word dispatched as a whole (hot words
compiled as constants: one scalar branch per element instead of a dozen)
against the general code.  Two libraries, launches alternating."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from bench import workload_rays, FIELD_FRACTIONS, BUNDLE_RADIUS
import digest_cases as dc

n = 10_000_000
D = os.path.dirname(_build.LIB)
libs = {"general": _build.LIB,
        "general, blocks aligned 32": os.path.join(D, "librt_mi355_alignallblocks_5.so"),
        "general, no-fallthru blocks aligned 64": os.path.join(D, "librt_mi355_alignallnofallthrublocks_6.so"),
        "simple kernel": os.path.join(D, "librt_mi355_simple.so")}


def steady(eng, clip, seconds=1.6):
    """Median launch time over the last two thirds of `seconds` of
    back-to-back launches (the power filter settles in ~0.5 s: variants must
    not be compared through short alternating blocks)."""
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, clip)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


def ab(name, make, clip=True, keep=None, rounds=2):
    g = {k: make(Engine(0, lib_path=p)) for k, p in libs.items()}
    for t in g.values():
        t.propagate(clip=clip, keep=keep)
    steady(g["general"].engine, clip, 2.)
    res = {k: [] for k in g}
    for rep in range(rounds):
        for k, t in (list(g.items()) if rep % 2 == 0 else
                     list(g.items())[::-1]):
            res[k].append(steady(t.engine, clip))
    same = all(np.array_equal(np.asarray(getattr(g["general"], a)[-1]),
                              np.asarray(getattr(t, a)[-1]), equal_nan=True)
               for a in "yut" for t in g.values())
    print(json.dumps({"what": name, **{k: v for k, v in res.items()},
                      "identical": bool(same)}), flush=True)
    del g


s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)


def host(eng):
    t = ra.GeometricTrace(s3, engine=eng)
    t.rays_given(y, u)
    return t


ab("C3 host-seeded", host)
ab("C3 host-seeded, unclipped", host, clip=False)

nf = len(FIELD_FRACTIONS)
pts = dc.disc_points(n//nf//64*64, 7)


def gen(eng):
    t = ra.GeometricTrace(s3, engine=eng)
    t.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS], pts,
                  P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
    return t


ab("C3 device-generated", gen)
s4 = ra.system_from_yaml(P.ASPHERE_PHONE)
y4, u4 = dc.bundle(n, .6, 10., 4)
y4[:, 1] -= .5*np.tan(np.radians(10.))


def asph(eng):
    t = ra.GeometricTrace(s4, engine=eng)
    t.rays_given(y4, u4)
    return t


ab("C4 asphere, default arithmetic", asph)
s2 = ra.system_from_yaml(P.COOKE % dict(air="air", sk16="SCHOTT-SK|N-SK16",
                                        f2="SCHOTT-F|N-F2"))
y2, u2 = dc.bundle(10**6, 5.5, 5., 0)


def cooke(eng):
    t = ra.GeometricTrace(s2, engine=eng)
    t.rays_given(y2, u2, l=[587.56e-9, 656.27e-9, 486.13e-9])
    return t


ab("C2 Cooke, 3 x 10^6 rays, one launch", cooke)
