"""(laboratory build, laboratory options key=value ... from argv) The per-allocation choice between two and four workgroups per CU
(rt_tuning): in one process, several contexts (= several allocations), each
timed with the choice switched off (two per CU), forced to four, and made by
the engine; what it chose and what it measured while choosing."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import rayopt_amd as ra
from rayopt_amd import prescriptions as P
from rayopt_amd.engine import Engine as _Engine
from rayopt_amd import _build
LAB = os.path.join(os.path.dirname(_build.LIB), 'librt_mi355_probes.so')
OPTS = [a.split('=') for a in sys.argv[1:]]
ROUND = ' '.join(sys.argv[1:])


def Engine(dev):
    e = _Engine(dev, lib_path=LAB)
    for k, v in OPTS:
        e.set_option(k, int(v))
    return e
from bench import workload_rays, FIELD_FRACTIONS, BUNDLE_RADIUS
import digest_cases as dc

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
nf = len(FIELD_FRACTIONS)
pts = dc.disc_points(n//nf//64*64, 7)


def steady(g, seconds=.9, clip=True):
    eng = g.engine
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            g.propagate(clip=clip)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


def host():
    g = ra.GeometricTrace(system, engine=Engine(0))
    g.rays_given(y, u)
    return g


def gen():
    g = ra.GeometricTrace(system, engine=Engine(0))
    g.rays_fields(np.c_[np.zeros(nf), FIELD_FRACTIONS], pts,
                  P.DOUBLE_GAUSS_PUPIL_Z, BUNDLE_RADIUS)
    return g


keep = []
first = host()
first.propagate(clip=True)
steady(first, 2.)
for kind, make in (("host-seeded", host), ("device-generated", gen)):
    for k in range(3):
        g = make()
        keep.append(g)
        eng = g.engine
        eng.set_option("tune_resident", 0)
        g.propagate(clip=True)
        two = steady(g)
        eng.set_option("resident_lds", 32768)
        four = steady(g)
        eng.set_option("resident_lds", -1)
        eng.set_option("tune_resident", 1)
        tuned = steady(g)
        st, lds, ms = eng.tuning()
        print(json.dumps({"options": ROUND, "kind": kind, "context": k, "two_per_cu_ms": two,
                          "four_per_cu_ms": four, "tuned_ms": tuned,
                          "tuning_state": st, "chosen_lds": lds,
                          "measured_while_choosing_ms": ms,
                          "Y_row1_address": hex(eng.device_ptr(1, 1))}), flush=True)
