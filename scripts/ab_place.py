"""Two BUILDS of the library side by side in one process, contexts alternating
(round 5): how the arrays come to lie behind their address range is the only
thing that differs between the round-4 build (pieces classified by pair tests
in a scratch range, an even mix of classes re-mapped) and this round's (pieces
mapped once, the store pattern measured, another range if it is slow).  Raw
ctypes on the calls both ABIs share; C2 (3 x 10^6 rays), C3 (10^7) and C3'
(per-ray directions) per build and turn, the settled launch time of each.

    python scripts/ab_place.py OLD.so NEW.so [turns]
"""
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import rayopt_amd as ra                                    # noqa: E402
from rayopt_amd import prescriptions as P                  # noqa: E402
from rayopt_amd.pack import pack_system                    # noqa: E402
import digest_cases as dc                                  # noqa: E402
from bench import workload_rays                            # noqa: E402

P_ = ctypes.c_void_p


def load(path):
    lib = ctypes.CDLL(path)
    lib.rt_last_error.restype = ctypes.c_char_p
    lib.rt_last_error.argtypes = [P_]
    lib.rt_create.argtypes = [ctypes.c_int, ctypes.POINTER(P_)]
    lib.rt_destroy.argtypes = [P_]
    lib.rt_upload_system_groups.argtypes = [P_, P_, ctypes.c_int, ctypes.c_int]
    lib.rt_set_rays_repeat.argtypes = [P_, P_, P_, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_int]
    lib.rt_trace.argtypes = [P_, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.rt_sync.argtypes = [P_]
    lib.rt_event_record.argtypes = [P_, ctypes.c_int]
    lib.rt_event_elapsed.argtypes = [P_, ctypes.c_int, ctypes.c_int,
                                     ctypes.POINTER(ctypes.c_double)]
    lib.rt_kernel_ms.argtypes = [P_, ctypes.POINTER(ctypes.c_double)]
    lib.rt_placement.argtypes = [P_, ctypes.POINTER(ctypes.c_int),
                                 ctypes.POINTER(ctypes.c_double)]
    lib.rt_download.argtypes = [P_, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                P_]
    lib.rt_set_option.argtypes = [P_, ctypes.c_char_p, ctypes.c_int]
    return lib


def check(lib, ctx, rc, what):
    if rc != 0:
        raise RuntimeError("%s: %s" % (what, lib.rt_last_error(ctx).decode()))


def one(lib, tables, y, u, copies, L, placed=1):
    ctx = P_()
    check(lib, None, lib.rt_create(0, ctypes.byref(ctx)), "create")
    check(lib, ctx, lib.rt_set_option(ctx, b"placement", placed), "option")
    t = np.ascontiguousarray(tables)
    check(lib, ctx, lib.rt_upload_system_groups(ctx, t.ctypes.data, L,
                                                len(tables)), "upload")
    check(lib, ctx, lib.rt_set_rays_repeat(ctx, y.ctypes.data, u.ctypes.data,
                                           len(y), copies, 0), "rays")
    check(lib, ctx, lib.rt_trace(ctx, 1, 0, 1), "trace")
    lib.rt_sync(ctx)
    ms = ctypes.c_double()
    lib.rt_kernel_ms(ctx, ctypes.byref(ms))
    per = max(1, min(10, int(40./max(ms.value, 1e-3))))
    t_end = time.perf_counter() + .3
    while time.perf_counter() < t_end:
        for _ in range(per):
            lib.rt_trace(ctx, 1, 0, 1)
        lib.rt_sync(ctx)
    ts = []
    t_end = time.perf_counter() + .4
    while time.perf_counter() < t_end or len(ts) < 5:
        lib.rt_event_record(ctx, 0)
        for _ in range(per):
            lib.rt_trace(ctx, 1, 0, 1)
        lib.rt_event_record(ctx, 1)
        lib.rt_event_elapsed(ctx, 0, 1, ctypes.byref(ms))
        ts.append(ms.value/per)
    info = (ctypes.c_int*16)()
    pm = (ctypes.c_double*16)()
    lib.rt_placement(ctx, info, pm)
    n = len(y)*copies
    img = np.empty((3, n))
    check(lib, ctx, lib.rt_download(ctx, 0, L - 1, L, img.ctypes.data), "down")
    lib.rt_destroy(ctx)
    return float(np.median(ts)), list(info), [round(v, 3) for v in pm], img


def main():
    old, new = load(sys.argv[1]), load(sys.argv[2])
    turns = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    s2 = ra.system_from_yaml(P.COOKE % dict(
        air="air", sk16="SCHOTT-SK|N-SK16", f2="SCHOTT-F|N-F2"))
    ls = [587.56e-9, 656.27e-9, 486.13e-9]
    t2 = np.stack([pack_system(s2, l, s2.refractive_index(l, 0))[0]
                   for l in ls])
    y2, u2 = dc.bundle(10**6, 5.5, 5., 0)
    s3 = ra.system_from_yaml(P.DOUBLE_GAUSS)
    l3 = s3.wavelengths[0]
    t3 = np.stack([pack_system(s3, l3, s3.refractive_index(l3, 0))[0]])
    y3, u3 = workload_rays(10_000_000, 0)
    y3p, u3p = workload_rays(10_000_000, 7)
    rng = np.random.default_rng(3)
    u3p[:, 0] += 1e-7*rng.standard_normal(len(u3p))
    u3p[:, 1] += 1e-7*rng.standard_normal(len(u3p))
    u3p[:, 2] = np.sqrt(1. - u3p[:, 0]**2 - u3p[:, 1]**2)
    shapes = (("C2", t2, y2, u2, 3, len(s2)), ("C3", t3, y3, u3, 1, len(s3)),
              ("C3'", t3, y3p, u3p, 1, len(s3)))
    want = {}
    for k in range(turns):
        for name, tab, y, u, copies, L in shapes:
            for tag, lib, placed in (("r04", old, 1), ("r05", new, 1),
                                     ("r04 hipMalloc", old, 0),
                                     ("r05 hipMalloc", new, 0)):
                ms, info, pm, img = one(lib, tab, np.ascontiguousarray(y),
                                        np.ascontiguousarray(u), copies, L,
                                        placed)
                same = np.array_equal(want.setdefault(name, img), img,
                                      equal_nan=True)
                print(json.dumps({"shape": name, "turn": k, "build": tag,
                                  "trace_ms": ms, "info": info, "ms": pm,
                                  "image_rows_equal": bool(same)}),
                      flush=True)


if __name__ == "__main__":
    main()
