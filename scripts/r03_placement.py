"""What is the per-process state that decides whether four workgroups per CU
are 2 % faster or 5 % slower than two (r03_full_i.py: stable within a process,
different between processes)?  In ONE process:
  part 1  six contexts of one library (same code, six sets of arrays)
  part 2  four more contexts, each created after a spacer allocation of an odd
          size that is kept (the arrays land elsewhere)
  part 3  four copies of the library file, one context each (same code at
          four load addresses)
each timed at two workgroups per CU (resident_lds 65536, the default), four
(32768) and no cap (0)."""
import json, os, shutil, sys, tempfile, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rayopt_amd as ra
from rayopt_amd import prescriptions as P, _build
from rayopt_amd.engine import Engine
from rayopt_amd._lib import RT_Y
from bench import workload_rays

n = 10_000_000
system = ra.system_from_yaml(P.DOUBLE_GAUSS)
y, u = workload_rays(n, 0)
LDS = (65536, 32768, 0)


def steady(eng, seconds=.9):
    t_end = time.time() + seconds
    ms = []
    while time.time() < t_end:
        eng.event_record(0)
        for _ in range(10):
            eng.trace(1, 0, True)
        eng.event_record(1)
        ms.append(eng.event_elapsed(0, 1)/10)
    return float(np.median(ms[len(ms)//3:]))


def make(lib=None):
    g = ra.GeometricTrace(system, engine=Engine(0, lib_path=lib))
    g.rays_given(y, u)
    g.propagate(clip=True)
    return g


def report(part, k, g, note=None):
    res = {}
    for lds in LDS:
        g.engine.set_option("resident_lds", lds)
        res[str(lds)] = steady(g.engine)
    g.engine.set_option("resident_lds", -1)
    addr = g.engine.device_ptr(RT_Y, 1)
    print(json.dumps({"part": part, "context": k, "note": note,
                      "steady_ms_by_resident_lds": res,
                      "Y_row1_address": hex(addr)}), flush=True)


first = make()
steady(first.engine, 2.)
keep = [first]
report(1, 0, first)
for k in range(1, 6):
    g = make()
    keep.append(g)
    report(1, k, g)
report(1, 0, first, "the first context again")
spacers = []
for k, size in enumerate(((1 << 30) + 4096, (3 << 30) + (1 << 20) + 8192,
                          (512 << 20) + 65536, (5 << 30) + (37 << 12))):
    sp = Engine(0)
    sp.scratch(size)
    spacers.append(sp)
    g = make()
    keep.append(g)
    report(2, k, g, "after a kept spacer of %d bytes" % size)
tmp = tempfile.mkdtemp(prefix="rt_place_")
for k in range(4):
    path = os.path.join(tmp, "librt_copy%d.so" % k)
    shutil.copy(_build.LIB, path)
    g = make(path)
    keep.append(g)
    report(3, k, g, "own copy of the library")
report(1, 0, first, "the first context, at the end")
shutil.rmtree(tmp, ignore_errors=True)
