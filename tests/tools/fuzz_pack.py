"""Stateful fuzz of the packer's row/table cache (rayopt_amd/pack.py).

A System is mutated step by step in every way a caller can reach -- attribute
assignments (distance, curvature, conic, radius, angles, direction, material,
aspheric lists), ``set_path``, and IN-PLACE edits of every array the elements
and materials hand out (``offset[k] += d``, ``rot_normal[...]``, aspheric and
dispersion coefficient arrays, ``GasFormula.b/c``) -- and after every step the
table the live System packs (through its caches) must be byte for byte the
table a cache-less deep copy packs from scratch.  The reference re-reads every
attribute on every ``propagate()`` (rayopt/system.py:459-464), so a cache that
survives any of these edits traces a stale geometry.

    python tests/tools/fuzz_pack.py 0 100 [steps per sequence]
"""
import copy
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import rayopt_amd as ra                                     # noqa: E402
from rayopt_amd import model                                # noqa: E402
from rayopt_amd.pack import pack_system                     # noqa: E402
from random_systems import random_prescription              # noqa: E402

MUTATIONS = ("distance", "curvature", "conic", "radius", "angles",
             "direction", "material", "aspherics_assign",
             "offset_inplace", "offset_setpath", "rot_inplace",
             "aspherics_inplace", "coeff_inplace", "gas_inplace",
             "material_attr", "nothing")


def scratch(system):
    """A deep copy without any of the packer's notes."""
    other = copy.deepcopy(system)
    other.__dict__.pop("_pack_table", None)
    for el in other:
        el.__dict__.pop("_pack_rows", None)
    return other


def packed_bytes(system, l, n0, start=1, stop=None):
    table, n = pack_system(system, l, n0, start, stop)
    return bytes(table.tobytes()), n.tobytes()


def named():
    from rayopt_amd import prescriptions as P
    import yaml
    texts = [P.COOKE % dict(air=1.0, sk16="1.62041/60.32",
                            f2="1.62004/36.37"),
             P.DOUBLE_GAUSS, P.TORTURE, P.ASPHERE_PHONE]
    return [yaml.safe_load(t) for t in texts]


NAMED = named()


def mutate(system, rng, kind):
    """Apply one mutation; returns a log string (or None if it did not
    apply to the element drawn)."""
    L = len(system)
    j = int(rng.integers(1, L))
    el = system[j]
    mat = getattr(el, "material", None)
    if kind == "distance":
        el.distance = el.distance*(1. + 1e-3*rng.normal())
    elif kind == "curvature" and hasattr(el, "curvature"):
        el.curvature = el.curvature*(1. + 1e-3*rng.normal()) + 1e-6
    elif kind == "conic" and hasattr(el, "conic"):
        el.conic = float(rng.uniform(-1.5, .2))
    elif kind == "radius" and np.isfinite(el.radius):
        el.radius *= 1. + 1e-2*rng.normal()
    elif kind == "angles":
        el.angles = rng.uniform(-.02, .02, 3)
    elif kind == "direction":
        d = rng.uniform(-.01, .01, 2)
        el.direction = (d[0], d[1], 1. if el.direction[2] >= 0 else -1.)
    elif kind == "material" and hasattr(el, "material"):
        el.material = model.ConstantIndex(float(rng.uniform(1., 1.9)))
    elif kind == "aspherics_assign" and hasattr(el, "aspherics"):
        el.aspherics = None if rng.random() < .3 else \
            [float(rng.normal()*1e-6) for _ in range(int(rng.integers(1, 5)))]
    elif kind == "offset_inplace":
        el.offset[int(rng.integers(3))] += 1e-2*rng.normal()
    elif kind == "offset_setpath":
        system.set_path((j, "offset", int(rng.integers(3))),
                        float(el.offset[0] + 1e-2*rng.normal()))
    elif kind == "rot_inplace" and el.rotated:
        a = 1e-3*rng.normal()
        c, s = np.cos(a), np.sin(a)
        el.rot_normal[...] = el.rot_normal @ np.array(
            ((c, s, 0.), (-s, c, 0.), (0., 0., 1.)))
    elif kind == "aspherics_inplace" and \
            getattr(el, "aspherics", None) is not None:
        k = int(rng.integers(len(el.aspherics)))
        el.aspherics[k] *= 1. + 1e-2*rng.normal()
    elif kind == "coeff_inplace" and isinstance(mat, model.DispersionGlass):
        mat.coefficients[int(rng.integers(len(mat.coefficients)))] *= \
            1. + 1e-4*rng.normal()
    elif kind == "gas_inplace" and isinstance(mat, model.GasFormula):
        (mat.b if rng.random() < .5 else mat.c)[0] *= 1. + 1e-4*rng.normal()
    elif kind == "material_attr" and isinstance(mat, model.ConstantIndex):
        mat.n = float(rng.uniform(1., 1.9))
    elif kind == "material_attr" and isinstance(mat, model.AbbeGlass):
        mat.v = float(rng.uniform(20., 70.))
    elif kind == "nothing":
        pass
    else:
        return None
    return "%s[%d]" % (kind, j)


def sequence(seed, nsteps, mirror=None):
    """``mirror(system, log)``: extra check run after every step (the live
    reference, when the caller has one)."""
    rng = np.random.default_rng(seed)
    p = copy.deepcopy(NAMED[(seed//2) % len(NAMED)]) if seed % 2 == 0 else \
        random_prescription(seed)
    system = ra.system_from_dict(copy.deepcopy(p))
    L = len(system)
    if seed % 5 == 0:       # a medium with coefficient arrays of its own
        system[1].material = model.DispersionGlass(
            "sellmeier_squared", [1.04, 6e-3, .23, 2e-2, 1.01, 103.])
    if seed % 7 == 0:
        system[0].material = model.GasFormula([5792105e-8, 167917e-8],
                                              [238.0185, 57.362])
    ls = list(system.wavelengths) + [system.wavelengths[0]*1.1]
    log = []
    for step in range(nsteps):
        what = mutate(system, rng, str(rng.choice(MUTATIONS)))
        if what is None:
            continue
        log.append(what)
        l = ls[int(rng.integers(len(ls)))]
        if rng.random() < .7:
            start, stop = 1, None
        else:
            start = int(rng.integers(1, L))
            stop = int(rng.integers(start, L + 1))
        n0 = system.refractive_index(l, start - 1)
        got = packed_bytes(system, l, n0, start, stop)
        want = packed_bytes(scratch(system), l, n0, start, stop)
        if got != want:
            raise AssertionError("stale packed table after: " +
                                 " | ".join(log[-8:]))
        if mirror is not None:
            mirror(system, log)
    return len(log)


def main():
    lo, hi = int(sys.argv[1]), int(sys.argv[2])
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    bad = steps = 0
    for seed in range(lo, hi):
        try:
            with np.errstate(all="ignore"):
                steps += sequence(seed, nsteps)
        except AssertionError as err:
            bad += 1
            print("FAIL seed %d: %s" % (seed, str(err)[:400]), flush=True)
    print("pack fuzz %d..%d: %d steps, %d failing sequences" % (
        lo, hi, steps, bad))
    return bad


if __name__ == "__main__":
    raise SystemExit(1 if main() else 0)
