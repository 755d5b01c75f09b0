"""The C ABI used directly: raw ctypes calls (no Python wrapper classes) and a
plain-C host program, on a real GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import rayopt_amd as ra
from rayopt_amd import _lib
from rayopt_amd.pack import pack_system
from oracle import trace_numpy as tn

from conftest import assert_parity, RTOL_SPHERICAL

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_raw_abi_call_sequence_and_errors():
    dll = _lib.load()
    ctx = ctypes.c_void_p()
    assert dll.rt_create(0, ctypes.byref(ctx)) == 0
    # call order is enforced with RT_ERR_STATE (-2), arguments with -1
    assert dll.rt_trace(ctx, 1, 0, 0) == -2
    assert b"upload" in dll.rt_last_error(ctx)
    assert dll.rt_reserve(ctx, 10) == -2
    system = ra.system_from_yaml(ra.prescriptions.COOKE %
                                 ra.prescriptions.COOKE_INDICES[587.56e-9])
    table, ns = pack_system(system, 587.56e-9, 1.0002771748755976)
    assert dll.rt_upload_system(ctx, None, 9) == -1
    assert dll.rt_upload_system(ctx, table.ctypes.data, 1) == -1
    assert dll.rt_upload_system(ctx, table.ctypes.data, len(table)) == 0
    assert dll.rt_nsurf(ctx) == 9
    y, u = ra.bundles.disc_bundle(5000, 5.5, 5., 2)
    ys, us = np.ascontiguousarray(y.T), np.ascontiguousarray(u.T)
    assert dll.rt_set_rays(ctx, ys.ctypes.data, us.ctypes.data, 5000, 7) == -1
    assert dll.rt_set_rays(ctx, ys.ctypes.data, us.ctypes.data, 5000,
                           _lib.LAYOUT_SOA) == 0
    assert dll.rt_nrays(ctx) == 5000
    if not os.environ.get("RT_MI355_BLOCK_RAYS"):   # (one block: 64-ray tiles)
        assert dll.rt_ld(ctx) == 5056
    blocks = (ctypes.c_int64*3)()
    assert dll.rt_blocks(ctx, blocks) == 0
    assert blocks[0]*blocks[1] == dll.rt_ld(ctx) >= 5000
    assert dll.rt_trace(ctx, 0, 0, 1) == -1
    assert dll.rt_trace(ctx, 1, 0, 1) == 0
    ms = ctypes.c_double()
    assert dll.rt_kernel_ms(ctx, ctypes.byref(ms)) == 0 and ms.value > 0
    want = tn.propagate(table, y, u, clip=True)
    for which, ref in zip((_lib.RT_Y, _lib.RT_U, _lib.RT_I), want[:3]):
        out = np.empty((8, 3, 5000))
        assert dll.rt_download(ctx, which, 1, 9, out.ctypes.data) == 0
        assert_parity(out.transpose(0, 2, 1), ref, RTOL_SPHERICAL, "raw")
    out = np.empty((8, 5000))
    assert dll.rt_download(ctx, _lib.RT_T, 1, 9, out.ctypes.data) == 0
    assert_parity(out, want[3], RTOL_SPHERICAL, "raw.t")
    assert dll.rt_download(ctx, _lib.RT_T, 5, 5, out.ctypes.data) == -2
    assert dll.rt_download(ctx, 9, 1, 2, out.ctypes.data) == -1
    col = np.empty((9, 3))
    assert dll.rt_download_ray(ctx, _lib.RT_Y, 17, col.ctypes.data) == 0
    assert_parity(col[1:][:, None, :], want[0][:, 17:18, :], RTOL_SPHERICAL,
                  "ray")
    assert dll.rt_download_ray(ctx, _lib.RT_Y, 5000, col.ctypes.data) == -2
    assert dll.rt_set_option(ctx, b"no_such_key", 1) == -1
    assert dll.rt_destroy(ctx) == 0
    assert dll.rt_destroy(None) == 0


def test_c_example(tmp_path):
    exe = tmp_path / "trace_singlet"
    subprocess.check_call([
        "gcc", "-O2", "-I" + os.path.join(ROOT, "include"),
        os.path.join(ROOT, "examples", "trace_singlet.c"),
        "-L" + os.path.join(ROOT, "rayopt_amd"), "-lrt_mi355", "-lm",
        "-Wl,-rpath," + os.path.join(ROOT, "rayopt_amd"),
        "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    out = subprocess.check_output([str(exe), "200001"], text=True)
    fields = out.split()
    assert fields[0] == "rays" and int(fields[1]) == 200001
    # the same fan through this package's model + the oracle
    n = 200001
    y = np.zeros((n, 3))
    y[:, 1] = -9.5 + 19.*np.arange(n)/(n - 1)
    u = np.zeros((n, 3))
    u[:, 2] = 1.
    system = ra.system_from_yaml(ra.prescriptions.SINGLET)
    table, _ = pack_system(system, 587.56e-9, 1.0)
    Y = tn.propagate(table, y, u, clip=True)[0]
    mid = n//2 + n//20
    assert float(fields[fields.index("y_image[mid]") + 1]) == \
        pytest.approx(Y[-1][mid, 1], rel=1e-9)


def test_failed_allocation_leaves_an_empty_context():
    """A batch that cannot be allocated (10^12 rays: 80 TB) is an error with
    the size in the message; the context holds no rays afterwards -- every
    entry that needs rows says so instead of touching a freed buffer -- and
    takes a new batch as if nothing had happened."""
    dll = _lib.load()
    ctx = ctypes.c_void_p()
    assert dll.rt_create(0, ctypes.byref(ctx)) == 0
    system = ra.system_from_yaml(ra.prescriptions.COOKE %
                                 ra.prescriptions.COOKE_INDICES[587.56e-9])
    table, ns = pack_system(system, 587.56e-9, 1.0002771748755976)
    assert dll.rt_upload_system(ctx, table.ctypes.data, len(table)) == 0
    y, u = ra.bundles.disc_bundle(3000, 5.5, 5., 2)
    ys, us = np.ascontiguousarray(y.T), np.ascontiguousarray(u.T)
    assert dll.rt_set_rays(ctx, ys.ctypes.data, us.ctypes.data, 3000,
                           _lib.LAYOUT_SOA) == 0
    assert dll.rt_trace(ctx, 1, 0, 1) == 0
    first = np.empty((3, 3000))
    assert dll.rt_download(ctx, _lib.RT_Y, 8, 9, first.ctypes.data) == 0

    assert dll.rt_reserve(ctx, 10**12) == -5            # RT_ERR_NOMEM
    assert b"GB failed" in dll.rt_last_error(ctx)
    assert dll.rt_nrays(ctx) == 0 and dll.rt_ld(ctx) == 0
    out = np.empty((3, 3000))
    assert dll.rt_trace(ctx, 1, 0, 1) == -2
    assert dll.rt_download(ctx, _lib.RT_Y, 8, 9, out.ctypes.data) == -2
    rms = ctypes.c_double()
    assert dll.rt_rms(ctx, 8, -1, ctypes.byref(rms)) == -2
    tiles = (ctypes.c_int64*7)()
    assert dll.rt_input_uniform(ctx, tiles) == 0 and not any(tiles)

    assert dll.rt_set_rays(ctx, ys.ctypes.data, us.ctypes.data, 3000,
                           _lib.LAYOUT_SOA) == 0
    assert dll.rt_trace(ctx, 1, 0, 1) == 0
    assert dll.rt_download(ctx, _lib.RT_Y, 8, 9, out.ctypes.data) == 0
    assert np.array_equal(out, first, equal_nan=True)
    assert dll.rt_destroy(ctx) == 0
