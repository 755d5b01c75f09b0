"""Randomised differential tests.

CPU (here, with /root/reference): oracle == unmodified reference on random
systems (pins the oracle far beyond the hand-written fixtures), and the
kernel arithmetic (host build of rt_math.h) agrees with the reference at the
contract tolerances.  GPU: the HIP engine against the oracle on the same
seeds.
"""
import copy

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn
from oracle import refshim

from conftest import (assert_parity, RTOL_SPHERICAL, RTOL_ASPHERE,
                      blas_follows_fma_chain)
from random_systems import random_prescription, random_rays

SEEDS = list(range(60))
# the numpy oracle's 3x3 products go through this host's BLAS: bit-identity
# through tilted elements is asserted where that BLAS sums as the kernel does
EXACT_TILTS = blas_follows_fma_chain()


def has_asphere(p):
    return any("aspherics" in e for e in p["elements"])


def tilted(p):
    return any("angles" in e or "direction" in e for e in p["elements"])


@pytest.mark.skipif(not refshim.available(), reason="no /root/reference")
@pytest.mark.parametrize("seed", SEEDS)
def test_oracle_and_kernel_math_vs_live_reference(seed, hostemu):
    ro = refshim.load()
    p = random_prescription(seed)
    n = 48 if has_asphere(p) else 400
    y, u = random_rays(seed, n, p)
    ref_sys = ro.System(**copy.deepcopy(p))
    mine = ra.system_from_dict(copy.deepcopy(p))
    for clip in (True, False):
        g = ro.GeometricTrace(ref_sys)
        g.rays_given(y, u)
        with np.errstate(all="ignore"):
            g.propagate(clip=clip)
        want = (g.y[1:], g.u[1:], g.i[1:], g.t[1:])
        # (1) oracle on the reference's own System object
        table, ns = pack_system(ref_sys, g.l, g.n[0])
        got = tn.propagate(table, y, u, clip=clip)
        assert np.array_equal(ns[1:], g.n[1:])
        for a, b in zip(got, want):
            assert np.array_equal(a, b, equal_nan=True), seed
        # (2) this package's host model packs the same system
        table2, _ = pack_system(mine, g.l, g.n[0])
        for f in table.dtype.names:
            np.testing.assert_allclose(table2[f], table[f], rtol=0,
                                       atol=1e-15)
        # (3) the kernel's arithmetic at the contract tolerance
        emu = hostemu(table2, y, u, 1, len(table2), clip, 1)
        rtol = RTOL_ASPHERE if has_asphere(p) else RTOL_SPHERICAL
        for a, b in zip(emu, want):
            assert_parity(a, b, rtol, "kernel math seed %d" % seed)
            if EXACT_TILTS or not tilted(p):
                # the reference's values bit for bit, through tilted elements
                # and iterated aspheres as well
                assert np.array_equal(a, b, equal_nan=True), seed


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_gpu_vs_oracle_random_systems(seed, arith):
    p = random_prescription(seed)
    if arith == "default" and not has_asphere(p):
        pytest.skip("no aspheric element: one arithmetic")
    system = ra.system_from_dict(copy.deepcopy(p))
    n = 20011
    y, u = random_rays(seed, n, p)
    for clip in (True, False):
        g = ra.GeometricTrace(system)
        g.rays_given(y, u)
        g.propagate(clip=clip)
        table, ns = pack_system(system, g.l, g.n[0])
        want = tn.propagate(table, y, u, clip=clip)
        rtol = RTOL_ASPHERE if has_asphere(p) else RTOL_SPHERICAL
        for rows, b in zip((g.y, g.u, g.i, g.t), want):
            assert_parity(np.asarray(rows[1:]), b, rtol, "seed %d" % seed)
            if (EXACT_TILTS or not tilted(p)) and arith == "exact":
                assert np.array_equal(np.asarray(rows[1:]), b,
                                      equal_nan=True), seed
        assert np.array_equal(g.n[1:], ns[1:])


ASPHERIC_SEEDS = [s for s in range(200) if has_asphere(random_prescription(s))]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [s for s in ASPHERIC_SEEDS if s >= len(SEEDS)])
def test_gpu_vs_oracle_every_aspheric_system(seed, arith):
    """The aspheric systems among seeds 60..199 as well, on both arithmetics:
    the exact one == the oracle, the shipped default within 1e-8 of it with
    identical NaN masks (a flipped convergence decision of the five-iterate
    Newton solve would show as a mask difference)."""
    test_gpu_vs_oracle_random_systems(seed, arith)


@pytest.mark.parametrize("seed", ASPHERIC_SEEDS[:30])
def test_fast_asphere_arithmetic_on_random_systems(seed, hostemu):
    """RT_F_FAST on the host build of the kernel arithmetic against the exact
    path of the same header, random aspheric systems (tilts, conics, mirrors,
    up to five terms): 1e-8 contract, identical NaN masks."""
    from rayopt_amd._lib import F_ASPH, F_FAST
    p = random_prescription(seed)
    system = ra.system_from_dict(copy.deepcopy(p))
    y, u = random_rays(seed, 600, p)
    table, _ = pack_system(system, 587.56e-9,
                           system.refractive_index(587.56e-9, 0))
    fast = table.copy()
    fast["flags"] = np.where(fast["flags"] & F_ASPH, fast["flags"] | F_FAST,
                             fast["flags"])
    for clip in (True, False):
        a = hostemu(table, y, u, 1, len(table), clip, 1)
        b = hostemu(fast, y, u, 1, len(table), clip, 1)
        for x, w in zip(b, a):
            assert_parity(x, w, RTOL_ASPHERE, "seed %d" % seed)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", ASPHERIC_SEEDS[:30])
def test_gpu_fast_asphere_and_compaction_on_random_systems(seed):
    """The opt-in kernel variants on random aspheric systems: fast arithmetic
    within the asphere contract of the oracle, and the compacting kernel
    bit-identical to the plain one (every row and the image row only)."""
    p = random_prescription(seed)
    system = ra.system_from_dict(copy.deepcopy(p))
    n = 20011
    y, u = random_rays(seed, n, p)
    table, ns = pack_system(system, 587.56e-9,
                            system.refractive_index(587.56e-9, 0))
    want = tn.propagate(table, y, u, clip=True)
    g = ra.GeometricTrace(system, fast_asphere=True)
    try:
        g.rays_given(y, u)
        g.propagate(clip=True)
        for rows, b in zip((g.y, g.u, g.i, g.t), want):
            assert_parity(np.asarray(rows[1:]), b, RTOL_ASPHERE,
                          "fast seed %d" % seed)
    finally:
        g.engine.set_option("fast_asphere", 0)
    g.propagate(clip=True)
    plain = [np.array(np.asarray(r[1:])) for r in (g.y, g.u, g.i, g.t)]
    g.engine.set_option("compact", 2)
    g.engine.set_option("compact_every", 1)
    try:
        g.propagate(clip=True)
        for rows, b in zip((g.y, g.u, g.i, g.t), plain):
            assert np.array_equal(np.asarray(rows[1:]), b, equal_nan=True), \
                seed
        g.engine.set_option("compact", 1)
        g.propagate(clip=True, keep=[-1])
        assert np.array_equal(np.asarray(g.y[-1]), plain[0][-1],
                              equal_nan=True), seed
    finally:
        g.engine.set_option("compact", 0)
        g.engine.set_option("compact_every", 4)
