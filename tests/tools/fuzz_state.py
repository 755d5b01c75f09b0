"""Stateful fuzz of the engine's bookkeeping on the GPU: the same random
sequence of calls -- seedings (host rays, weights, device-built bundles),
partial and full traces with and without clipping and kept-row subsets, reads
of single rows in random order, changes of the System between traces, kernel
options flipped on the device side only -- is applied to a GeometricTrace on
the device and to one on the numpy engine double, and after every step every
row the double holds must be bit for bit what the device holds.

What it is after: rows served instead of stored (i from u, u from i), rows
detached before their source is overwritten, keep masks, host-side row caches
invalidated at the right time, a generated batch rebuilt instead of read.

    python tests/tools/fuzz_state.py 0 200 [ops per sequence]
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rayopt_amd as ra
from fake_engine import OracleEngine
from random_systems import random_prescription, random_rays
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fuzz_pack

# (not 1: numpy hands a (1,3) @ (3,3) product to another BLAS routine than
# an (N,3) one, and the double -- like the reference -- then rounds a tilted
# element's rotation differently for that single ray)
SIZES = (2, 63, 64, 65, 777, 4099)
OPTIONS = (("alias_i", (0, 1)), ("regenerate", (0, 1)),
           ("fuse_generate", (0, 1)), ("compact", (0, 1, 2)),
           ("compact_every", (1, 2, 3, 4)), ("uniform_input", (0, 1)),
           ("resident_lds", (-1, 0, 65536)), ("range_shortcuts", (0, 1)))
# The double is the reference's arithmetic: with exact_asphere=1 the device
# holds its bits.  RT_FUZZ_ARITH=default runs the SHIPPED arithmetic for even
# aspheres instead: aspheric systems are then compared at the 1e-8 contract
# with identical NaN masks, everything else still bit for bit.
ARITH = os.environ.get("RT_FUZZ_ARITH", "exact")
DEFAULTS = dict(alias_i=1, regenerate=1, fuse_generate=1, compact=0,
                compact_every=4, exact_asphere=int(ARITH == "exact"),
                uniform_input=1, resident_lds=-1, range_shortcuts=1)
LOOSE = [False]     # this sequence's system is aspheric, default arithmetic


def same(x, w):
    if not LOOSE[0]:
        return np.array_equal(x, w, equal_nan=True)
    x, w = np.asarray(x, dtype=float), np.asarray(w, dtype=float)
    if x.shape != w.shape or not np.array_equal(np.isnan(x), np.isnan(w)):
        return False
    fin = np.isfinite(w)
    if not np.array_equal(x[~fin & ~np.isnan(w)], w[~fin & ~np.isnan(w)]):
        return False
    if not fin.any():
        return True
    scale = np.abs(w[fin]).max()
    return bool((np.abs(x[fin] - w[fin]) <=
                 1e-8*np.maximum(np.abs(w[fin]), scale)).all())


def compare(dev, cpu, log):
    L = cpu.length
    for name in "yuit":
        a, b = getattr(dev, name), getattr(cpu, name)
        for j in range(L):
            if not cpu.engine.valid[j]:
                continue
            x, w = np.asarray(a[j]), np.asarray(b[j])
            if not same(x, w):
                raise AssertionError("%s[%d] differs after: %s" % (
                    name, j, " | ".join(log[-10:])))
    held = np.asarray(cpu.engine.valid, dtype=bool)
    if not np.array_equal(np.asarray(dev.n)[..., held],
                          np.asarray(cpu.n)[..., held], equal_nan=True):
        raise AssertionError("n differs after: " + " | ".join(log[-6:]))


def named_prescriptions():
    """Designs with a field of view (the random ones image an axial point):
    aiming, generated bundles of several fields, dispersion."""
    import yaml
    from rayopt_amd import prescriptions as P
    texts = [P.COOKE % dict(air=1.0, sk16="1.62041/60.32",
                            f2="1.62004/36.37"),
             P.DOUBLE_GAUSS, P.TORTURE,
             P.ASPHERE_PHONE.replace("pupil: {radius: 0.6}",
                                     "pupil: {radius: 0.6, aim: True}")]
    return [yaml.safe_load(t) for t in texts]


NAMED = named_prescriptions()
PATTERNS = ("meridional", "sagittal", "cross", "tee", "square", "triangular",
            "hexapolar")


def sequence(seed, nops):
    rng = np.random.default_rng(seed)
    designed = seed % 3 == 0
    p = copy.deepcopy(NAMED[(seed//3) % len(NAMED)]) if designed else \
        random_prescription(seed)
    system = ra.system_from_dict(copy.deepcopy(p))
    LOOSE[0] = ARITH != "exact" and any(
        getattr(e, "aspherics", None) is not None for e in system)
    L = len(system)
    dev = ra.GeometricTrace(system)
    cpu = ra.GeometricTrace(system, engine=OracleEngine())
    eng = dev.engine
    for k, v in DEFAULTS.items():
        eng.set_option(k, v)
    log, seeded, grouped, last_seed = [], False, False, None
    try:
        for _ in range(nops):
            op = rng.choice(["given", "fields", "prop", "prop", "prop",
                             "read", "opt", "mutate", "upload", "reduce",
                             "groups", "variants"] +
                            (["points", "points"] if designed else []))
            if not seeded and op not in ("given", "fields", "groups",
                                         "variants", "points"):
                op = "given"
            if op == "given":
                n = int(rng.choice(SIZES))
                y, u = random_rays(int(rng.integers(1 << 30)), n, p)
                special = rng.random() < .4
                if special:
                    # a collimated bundle, or one from a point: components
                    # uniform across 64-ray tiles are fetched once per tile
                    if rng.random() < .5:
                        u[:] = u[0]
                    else:
                        y[:] = y[0]
                    k = int(rng.integers(n))    # ... but for one ray
                    u[k], y[k] = u[(k + 1) % n]*1., y[(k + 1) % n] + 1e-3
                w = None
                if rng.random() < .3:
                    w = rng.random(n)
                    w /= w.sum()
                for t in (dev, cpu):
                    t.rays_given(y, u, w=w)
                # (refocus of a parallel bundle is 0/0: not compared)
                seeded, grouped = True, False
                last_seed = "given, parallel or point" if special else "given"
                log.append("given n=%d w=%s special=%s" % (n, w is not None,
                                                           special))
            elif op == "points":
                # pattern -> aiming kernel (or first-order pupil) ->
                # generation -> trace, several fields, maybe all wavelengths
                nf = int(rng.integers(1, 5))
                fields = rng.uniform(-1, 1, (nf, 2))*.9
                ls = system.wavelengths
                kw = dict(nrays=int(rng.choice((12, 40, 90))),  # > 1 ray
                          distribution=str(rng.choice(PATTERNS)),
                          clip=bool(rng.random() < .5),
                          aim=[True, False, None][int(rng.integers(3))],
                          rim=bool(rng.random() < .3),
                          wavelength=list(ls) if len(ls) > 1 and
                          rng.random() < .5 else None)
                failed = []
                for t in (dev, cpu):
                    try:
                        t.rays_points(fields, **kw)
                        failed.append(None)
                    except ValueError as err:   # a field that cannot be aimed
                        failed.append(str(err))
                assert failed[0] == failed[1], \
                    "aiming: device %r, double %r" % tuple(failed)
                if failed[0] is not None:
                    # both refused in the same words; start over
                    seeded = False
                    log.append("points refused")
                    continue
                seeded, grouped = True, bool(kw["wavelength"])
                last_seed = "points"
                log.append("points %d fields %s" % (nf, kw))
            elif op == "groups":
                # the same rays at two wavelengths: two ray groups, one
                # surface table each, one launch
                m = int(rng.choice((64, 128, 4096)))
                y, u = random_rays(int(rng.integers(1 << 30)), m, p)
                l0 = system.wavelengths[0]
                for t in (dev, cpu):
                    t.rays_given(y, u, l=[l0, l0*1.07])
                seeded = grouped = True
                last_seed = "groups"
                log.append("given groups 2x%d" % m)
            elif op == "variants":
                # the same rays through two variants of the system
                m = int(rng.choice((5, 64, 777)))
                y, u = random_rays(int(rng.integers(1 << 30)), m, p)
                other = copy.deepcopy(system)
                k = int(rng.integers(1, L))
                if hasattr(other[k], "curvature"):
                    other[k].curvature *= 1.003
                other[k].distance = other[k].distance*1.001
                for t in (dev, cpu):
                    t.rays_variants(y, u, [system, other])
                seeded = grouped = True
                last_seed = "variants"
                log.append("given variants 2x%d" % m)
            elif op == "fields":
                nf = int(rng.integers(1, 5))
                m = int(rng.choice((64, 200, 333)))
                if (nf*m) % 1 == 0:
                    fields = rng.uniform(-1, 1, (nf, 2))
                    yp = rng.uniform(-.6, .6, (m, 2))
                    rad = min(float(e.radius) for e in system[1:-1]
                              if np.isfinite(e.radius))
                    for t in (dev, cpu):
                        t.rays_fields(fields, yp, 40., .5*rad)
                    seeded, grouped, last_seed = True, False, "fields"
                    log.append("fields %dx%d" % (nf, m))
            elif op == "prop":
                valid = [j for j in range(L - 1) if cpu.engine.valid[j]]
                start = int(rng.choice(valid)) + 1
                stop = None if rng.random() < .5 else \
                    int(rng.integers(start, L + 1))
                clip = bool(rng.random() < .5)
                keep = None
                if rng.random() < .3:
                    keep = sorted(set(int(x) for x in rng.integers(
                        0, L, int(rng.integers(1, L)))))
                # a step traced in pieces (rt_trace_chunk: what a multi-GPU
                # job overlaps its gather with) is the same step
                chunks = 1 if grouped else int(rng.choice((1, 1, 2, 3, 5)))
                for t in (dev, cpu):
                    t.propagate(start=start, stop=stop, clip=clip, keep=keep,
                                chunks=chunks)
                log.append("prop %d:%s clip=%s keep=%s chunks=%d" % (
                    start, stop, clip, keep, chunks))
            elif op == "read":
                j = int(rng.integers(0, L))
                name = "yuit"[int(rng.integers(4))]
                if cpu.engine.valid[j]:
                    x = np.asarray(getattr(dev, name)[j])
                    w = np.asarray(getattr(cpu, name)[j])
                    assert same(x, w), \
                        "read %s[%d] after: %s" % (name, j,
                                                   " | ".join(log[-6:]))
                log.append("read %s[%d]" % (name, j))
                continue
            elif op == "opt":
                key, values = OPTIONS[int(rng.integers(len(OPTIONS)))]
                value = int(rng.choice(values))
                eng.set_option(key, value)
                log.append("%s=%d" % (key, value))
                continue
            elif op == "reduce":
                # the device reductions read rows through the same aliases
                j = int(rng.integers(0, L))
                if cpu.engine.valid[j] and (j == 0 or cpu.engine.valid[j - 1]):
                    # (refocus is a ratio of two sums that both vanish for
                    # the parallel bundles `fields` makes of these systems'
                    # axial object point: rounding noise over rounding noise)
                    kind = rng.choice(["rms", "rms_ref", "rmax"] + (
                        ["refocus"] if last_seed == "given" else []))
                    if kind == "rms":
                        a, b = dev.rms(j), cpu.rms(j)
                    elif kind == "rms_ref":
                        r = int(rng.integers(cpu.nrays))
                        a, b = dev.rms(j, ref=r), cpu.rms(j, ref=r)
                    elif kind == "rmax":
                        a, b = dev.engine.row_rmax(j), cpu.engine.row_rmax(j)
                    else:
                        a = dev.engine.refocus_shift(j)
                        b = cpu.engine.refocus_shift(j)
                    ok = (np.isnan(a) and np.isnan(b)) or \
                        abs(a - b) <= 1e-8*max(1., abs(b)) or \
                        (not np.isfinite(b) and not np.isfinite(a))
                    assert ok, "%s(%d) %r != %r after: %s" % (
                        kind, j, a, b, " | ".join(log[-10:]))
                    log.append("%s(%d)" % (kind, j))
                continue
            elif op == "upload":
                # the caller's own data into one row of one array: rows that
                # were being served from it must keep what they showed
                j = int(rng.integers(0, L))
                if not cpu.engine.valid[j]:
                    continue
                which = int(rng.integers(4))
                n = cpu.nrays
                data = rng.normal(size=(1 if which == 3 else 3, n))
                if which == 3:
                    data = data[0]
                for t in (dev, cpu):
                    t.engine.upload_row(which, j, data)
                    for rows in (t.y, t.u, t.i, t.t):
                        rows.invalidate(0, L)
                log.append("upload %s[%d]" % ("yuit"[which], j))
            elif op == "mutate":
                j = int(rng.integers(1, L))
                el = system[j]
                what = rng.choice(["distance", "curvature", "radius",
                                   "offset_inplace", "rot_inplace",
                                   "aspherics_inplace"])
                if what == "curvature" and hasattr(el, "curvature"):
                    el.curvature *= 1. + 1e-3*rng.normal()
                elif what == "radius" and np.isfinite(el.radius):
                    el.radius *= 1. + 1e-2*rng.normal()
                elif what == "offset_inplace":
                    # arrays are handed out mutable, as the reference hands
                    # them out (rayopt/system.py:461 re-reads e.offset)
                    el.offset[int(rng.integers(3))] += 1e-3*rng.normal()
                elif what == "rot_inplace" and el.rotated:
                    a = 1e-3*rng.normal()
                    c, s = np.cos(a), np.sin(a)
                    el.rot_normal[...] = el.rot_normal @ np.array(
                        ((c, s, 0.), (-s, c, 0.), (0., 0., 1.)))
                elif what == "aspherics_inplace" and \
                        getattr(el, "aspherics", None) is not None:
                    el.aspherics[0] *= 1. + 1e-3*rng.normal()
                else:
                    what = "distance"
                    el.distance = el.distance*(1. + 1e-3*rng.normal())
                # the table the next trace packs through its caches must be
                # the one a cache-less copy packs (tests/tools/fuzz_pack.py)
                l0 = system.wavelengths[0]
                n00 = system.refractive_index(l0, 0)
                assert fuzz_pack.packed_bytes(system, l0, n00) == \
                    fuzz_pack.packed_bytes(fuzz_pack.scratch(system), l0,
                                           n00), \
                    "stale packed table after: " + " | ".join(
                        log[-6:] + ["mutate %d %s" % (j, what)])
                log.append("mutate %d %s" % (j, what))
                continue
            compare(dev, cpu, log)
    finally:
        for k, v in DEFAULTS.items():
            eng.set_option(k, v)
    return len(log)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    nops = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    bad = steps = 0
    for seed in range(lo, hi):
        try:
            with np.errstate(all="ignore"):
                steps += sequence(seed, nops)
        except AssertionError as err:
            bad += 1
            print("FAIL seed %d: %s" % (seed, str(err)[:400]), flush=True)
        except Exception as err:
            bad += 1
            print("ERROR seed %d: %r" % (seed, err), flush=True)
    print("state fuzz %d..%d: %d steps, %d failing sequences" % (
        lo, hi, steps, bad))


if __name__ == "__main__":
    main()
