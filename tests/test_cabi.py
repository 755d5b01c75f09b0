"""The C-ABI shared library: loads, exports every symbol the header declares,
agrees on the struct layout, and refuses to work without a GPU (no CPU
fallback).  No compute is issued here."""
import ctypes
import os
import re

import numpy as np
import pytest

from rayopt_amd import _lib, _build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "rt_mi355.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def dll():
    _build.build()
    return _lib.load()


def test_header_and_binding_agree(dll):
    names = declared_symbols()
    assert len(names) >= 25
    assert set(names) == set(_lib.SIGNATURES), \
        set(names) ^ set(_lib.SIGNATURES)


def test_every_declared_symbol_is_exported(dll):
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(raw, name), name


def test_shipped_library_carries_no_laboratory(dll):
    """The library exports what the header declares and nothing of a
    laboratory (the probes and rejected kernel variants of rounds 2-4 are in
    the repository's history, not in the product), and its kernel takes what
    the trace needs and nothing else."""
    import subprocess
    syms = subprocess.check_output(["nm", "-D", "--defined-only",
                                    _lib.LIB_PATH], text=True)
    assert "probe" not in syms and "_lab_" not in syms
    exported = {line.split()[-1] for line in syms.splitlines()
                if " T " in line and line.split()[-1].startswith("rt_")}
    assert exported == set(declared_symbols()), \
        exported ^ set(declared_symbols())
    # the trace kernel's signature: table, start, stop, clip, layout, ld,
    # group_rays, nsurf, ngroups, the tile notes of row 0 -- and nothing else
    # (two instantiations: tables with an aspheric element, and the lean
    # kernel without the Newton solves for tables that have none)
    for asph in "01":
        assert ("_Z15rt_trace_kernelILb%sEEvPK10rt_surfaceiii6rt_layllii8"
                "rt_tiles\n" % asph) in syms + "\n"


def test_struct_layout(dll):
    assert dll.rt_abi_version() == _lib.RT_ABI_VERSION == 5
    assert int(re.search(r"#define RT_ABI_VERSION (\d+)", open(HEADER).read()
                         ).group(1)) == 5
    assert dll.rt_sizeof_surface() == _lib.SURFACE_DTYPE.itemsize == 352
    text = open(HEADER).read()
    assert int(re.search(r"#define RT_MAX_ASPH (\d+)", text).group(1)) == \
        _lib.RT_MAX_ASPH
    assert int(re.search(r"#define RT_MAX_SURFACES (\d+)", text).group(1)) \
        == _lib.RT_MAX_SURFACES
    for flag in ("ROTATED", "CURVED", "CONIC", "ASPH", "ALT", "REFRACT",
                 "MIRROR"):
        val = int(re.search(r"#define RT_F_%s\s+0x([0-9a-f]+)u" % flag,
                            text).group(1), 16)
        assert val == getattr(_lib, "F_" + flag)


def test_stale_library_is_refused(dll, monkeypatch):
    """A library built from other sources (another ABI version) is refused at
    load instead of being called with the wrong struct layouts."""
    monkeypatch.setattr(_lib, "RT_ABI_VERSION", _lib.RT_ABI_VERSION + 1)
    monkeypatch.setattr(_lib, "_libs", {})
    with pytest.raises(_lib.EngineError, match="ABI version"):
        _lib.load()


def test_no_cpu_fallback_without_gpu(dll):
    count = ctypes.c_int(-1)
    rc = dll.rt_device_count(ctypes.byref(count))
    if rc == 0 and count.value > 0:
        pytest.skip("a GPU is visible")
    import rayopt_amd as ra
    with pytest.raises(ra.EngineError, match="rt_create"):
        ra.Engine()
    g = ra.GeometricTrace(ra.system_from_yaml(ra.prescriptions.SINGLET))
    with pytest.raises(ra.EngineError):
        g.rays_given(np.zeros((2, 3)), np.array([[0, 0, 1.]]))
    with pytest.raises(ra.EngineError):
        list(ra.system_from_yaml(ra.prescriptions.SINGLET).propagate(
            np.zeros((2, 3)), np.array([[0, 0, 1.]]), 1., 5e-7))
    assert b"ROCm" in dll.rt_last_error(None) or dll.rt_last_error(None)


def test_product_does_not_import_oracle():
    """oracle/ is test infrastructure: nothing under rayopt_amd/ may
    reference it."""
    pkg = os.path.join(ROOT, "rayopt_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", text,
                                     flags=re.M), f
                assert "trace_numpy" not in text, f


def test_chunk_bounds_tile_any_batch(dll):
    """rt_chunk_bounds (host arithmetic, no GPU): pieces of whole 256-ray
    workgroups, contiguous, covering [0, n) once -- what rt_trace_chunk traces
    and what every rank computes for every other rank in rt_gather_chunk."""
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    for n in (0, 1, 63, 256, 257, 4097, 10**7, 12_500_000, 2**33 + 5):
        for q in (1, 2, 3, 4, 7, 8, 64):
            edge = 0
            for k in range(q):
                assert dll.rt_chunk_bounds(n, k, q, ctypes.byref(lo),
                                           ctypes.byref(hi)) == 0
                assert lo.value == edge and lo.value <= hi.value <= n
                assert lo.value % 256 == 0 or lo.value == n
                edge = hi.value
            assert edge == n
    assert dll.rt_chunk_bounds(10, 3, 3, ctypes.byref(lo),
                               ctypes.byref(hi)) != 0
    assert dll.rt_chunk_bounds(10, 0, 0, ctypes.byref(lo),
                               ctypes.byref(hi)) != 0
