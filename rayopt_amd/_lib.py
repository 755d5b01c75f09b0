"""ctypes binding of librt_mi355.so (C ABI: include/rt_mi355.h).

There is no CPU fallback: if the library is missing or no MI355X is visible
the product raises.  ``SURFACE_DTYPE`` mirrors ``struct rt_surface`` field for
field; the layout is verified against ``rt_sizeof_surface()`` at load time.
"""
import ctypes
import os

import numpy as np

RT_MAX_ASPH = 10
RT_MAX_SURFACES = 256

F_ROTATED, F_CURVED, F_CONIC, F_ASPH, F_ALT, F_REFRACT, F_MIRROR = (
    0x01, 0x02, 0x04, 0x08, 0x10, 0x20, 0x40)
F_FAST = 0x400          # set by the library (default; "exact_asphere" clears)
RT_Y, RT_U, RT_I, RT_T = 0, 1, 2, 3
RT_ABI_VERSION = 5      # include/rt_mi355.h
RT_OPD_STATS = 8        # doubles per bundle of rt_opd_stats
LAYOUT_SOA, LAYOUT_AOS = 0, 1

SURFACE_DTYPE = np.dtype([
    ("c", "f8"), ("k", "f8"), ("kw", "f8"), ("kc2", "f8"),
    ("radius2", "f8"),
    ("mu", "f8"), ("muf", "f8"), ("smu", "f8"), ("mu2m1", "f8"),
    ("n0", "f8"),
    ("offset", "f8", (3,)),
    ("rot", "f8", (9,)),
    ("asph", "f8", (RT_MAX_ASPH,)),
    ("dasph", "f8", (RT_MAX_ASPH,)),
    ("nasph", "i4"), ("flags", "u4"),
    ("rc", "f8"),           # the library's (device-side 1/c), callers leave 0
], align=True)

OPD_ARGS_DTYPE = np.dtype([
    ("nrows", "i4"), ("after", "i4"), ("image", "i4"), ("finite", "i4"),
    ("rot_after", "i4"), ("rot_image", "i4"), ("ref", "i8"),
    ("n0", "f8"), ("n_after", "f8"), ("radius", "f8"), ("lscale", "f8"),
    ("shift", "f8", (3,)), ("r_after", "f8", (9,)), ("r_image", "f8", (9,)),
], align=True)

FIELD_DTYPE = np.dtype([
    ("finite", "i4"), ("flip", "i4"), ("am", "f8"), ("z", "f8"),
    ("base", "f8", (3,)), ("u", "f8", (3,)), ("s", "f8", (3,)),
    ("m", "f8", (3,)),
], align=True)

AIM_SEED_DTYPE = np.dtype([
    ("finite", "i4"), ("telecentric", "i4"), ("group", "i4"), ("pad_", "i4"),
    ("yo", "f8", (2,)), ("dir", "f8", (3,)), ("point", "f8", (3,)),
    ("z0", "f8"), ("a0", "f8"),
], align=True)

AIM_ARGS_DTYPE = np.dtype([
    ("stop", "i4"), ("rim", "i4"), ("maxiter", "i4"), ("no_chief", "i4"),
    ("tol", "f8"),
], align=True)

LIB_PATH = os.environ.get("RT_MI355_LIB") or os.path.join(
    os.path.dirname(os.path.abspath(__file__)), "librt_mi355.so")

_c_double_p = ctypes.POINTER(ctypes.c_double)
_c_int64_p = ctypes.POINTER(ctypes.c_int64)
_ctx = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/rt_mi355.h declares
SIGNATURES = {
    "rt_abi_version": (ctypes.c_int, []),
    "rt_sizeof_surface": (ctypes.c_int, []),
    "rt_sizeof_opd_args": (ctypes.c_int, []),
    "rt_device_count": (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    "rt_create": (ctypes.c_int, [ctypes.c_int, ctypes.POINTER(_ctx)]),
    "rt_destroy": (ctypes.c_int, [_ctx]),
    "rt_last_error": (ctypes.c_char_p, [_ctx]),
    "rt_upload_system": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int]),
    "rt_upload_system_groups": (ctypes.c_int, [_ctx, ctypes.c_void_p,
                                               ctypes.c_int, ctypes.c_int]),
    "rt_set_rays_repeat": (ctypes.c_int, [_ctx, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_int]),
    "rt_reserve": (ctypes.c_int, [_ctx, ctypes.c_int64]),
    "rt_nrays": (ctypes.c_int64, [_ctx]),
    "rt_ld": (ctypes.c_int64, [_ctx]),
    "rt_blocks": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_int64)]),
    "rt_nsurf": (ctypes.c_int, [_ctx]),
    "rt_set_rays": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_void_p,
                                   ctypes.c_int64, ctypes.c_int]),
    "rt_set_rays_device": (ctypes.c_int, [_ctx, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_int64,
                                          ctypes.c_int]),
    "rt_sizeof_field": (ctypes.c_int, []),
    "rt_generate_rays": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int64]),
    "rt_upload_row": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_void_p]),
    "rt_trace": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                ctypes.c_int]),
    "rt_trace_chunk": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int]),
    "rt_chunk_bounds": (ctypes.c_int, [ctypes.c_int64, ctypes.c_int,
                                       ctypes.c_int, _c_int64_p, _c_int64_p]),
    "rt_set_keep_rows": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int]),
    "rt_sync": (ctypes.c_int, [_ctx]),
    "rt_kernel_ms": (ctypes.c_int, [_ctx, _c_double_p]),
    "rt_event_record": (ctypes.c_int, [_ctx, ctypes.c_int]),
    "rt_event_elapsed": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                        _c_double_p]),
    "rt_set_option": (ctypes.c_int, [_ctx, ctypes.c_char_p, ctypes.c_int]),
    "rt_download": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_void_p]),
    "rt_newton_census": (ctypes.c_int, [_ctx, ctypes.c_int,
                                        ctypes.POINTER(ctypes.c_uint64)]),
    "rt_download_xy": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_void_p]),
    "rt_download_ray": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_void_p]),
    "rt_download_rays": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_void_p]),
    "rt_set_weights": (ctypes.c_int, [_ctx, ctypes.c_void_p]),
    "rt_rms": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int64,
                              _c_double_p]),
    "rt_refocus_shift": (ctypes.c_int, [_ctx, ctypes.c_int, _c_double_p]),
    "rt_row_rmax": (ctypes.c_int, [_ctx, ctypes.c_int, _c_double_p]),
    "rt_sizeof_aim_seed": (ctypes.c_int, []),
    "rt_sizeof_aim_args": (ctypes.c_int, []),
    "rt_aim_pupil": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p]),
    "rt_spot_stats": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_void_p]),
    "rt_row_stats": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int64,
                                    ctypes.c_int, ctypes.c_int64,
                                    ctypes.c_void_p]),
    "rt_opd_rays": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_void_p]),
    "rt_opd_stats": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_int, ctypes.c_int,
                                    ctypes.c_void_p]),
    "rt_opd_device": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_void_p),
                                     ctypes.POINTER(ctypes.c_int64)]),
    "rt_device_ptr": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_void_p)]),
    "rt_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
    "rt_comm_init": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_int]),
    "rt_comm_destroy": (ctypes.c_int, [_ctx]),
    "rt_gather_final": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                       _c_int64_p, ctypes.c_int,
                                       ctypes.c_void_p]),
    "rt_gather_chunk": (ctypes.c_int, [_ctx, ctypes.c_int, ctypes.c_int,
                                       _c_int64_p, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_int,
                                       ctypes.c_int]),
    "rt_gather_ms": (ctypes.c_int, [_ctx, _c_double_p, _c_double_p]),
    "rt_comm_sync": (ctypes.c_int, [_ctx]),
    "rt_input_uniform": (ctypes.c_int, [_ctx, _c_int64_p]),
    "rt_input_completed": (ctypes.c_int, [_ctx, _c_int64_p]),
    "rt_comm_info": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_int),
                                    ctypes.c_int]),
    "rt_placement": (ctypes.c_int, [_ctx, ctypes.POINTER(ctypes.c_int),
                                    ctypes.POINTER(ctypes.c_double)]),
    "rt_selftest_arith": (ctypes.c_int, [_ctx, ctypes.c_uint64,
                                         ctypes.c_int64, ctypes.c_int,
                                         ctypes.POINTER(ctypes.c_uint64)]),
    "rt_scratch": (ctypes.c_int, [_ctx, ctypes.c_int64,
                                  ctypes.POINTER(ctypes.c_void_p)]),
    "rt_copy_to_host": (ctypes.c_int, [_ctx, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_int64]),
}


_libs = {}


class EngineError(RuntimeError):
    pass


def load(path=None):
    """Load librt_mi355.so (or the library at ``path``: the laboratory build
    next to the shipped one, for A/B measurements in one process); raises
    EngineError when it has not been built."""
    path = os.path.abspath(path or LIB_PATH)
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise EngineError(
            "%s not found: build it with `python -m rayopt_amd._build` "
            "(hipcc, gfx950). There is no CPU fallback." % path)
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.rt_abi_version() != RT_ABI_VERSION:
        raise EngineError(
            "%s speaks ABI version %d, this package %d: rebuild it "
            "(python -m rayopt_amd._build)" % (path, lib.rt_abi_version(),
                                               RT_ABI_VERSION))
    if lib.rt_sizeof_surface() != SURFACE_DTYPE.itemsize:
        raise EngineError("struct rt_surface is %d bytes in the library but "
                          "%d in SURFACE_DTYPE" % (lib.rt_sizeof_surface(),
                                                   SURFACE_DTYPE.itemsize))
    if lib.rt_sizeof_field() != FIELD_DTYPE.itemsize:
        raise EngineError("struct rt_field layout mismatch")
    if lib.rt_sizeof_opd_args() != OPD_ARGS_DTYPE.itemsize:
        raise EngineError("struct rt_opd_args layout mismatch")
    if lib.rt_sizeof_aim_seed() != AIM_SEED_DTYPE.itemsize or \
            lib.rt_sizeof_aim_args() != AIM_ARGS_DTYPE.itemsize:
        raise EngineError("struct rt_aim_seed / rt_aim_args layout mismatch")
    _libs[path] = lib
    return lib
