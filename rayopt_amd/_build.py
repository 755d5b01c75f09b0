"""Build the HIP engine (gfx950) in-tree with hipcc.

``librt_mi355.so`` -- the product: rt_engine.hip (contexts, tables, seeding,
the trace), rt_consumers.hip (aiming, rms / refocus / spot statistics / opd),
rt_comm.hip (RCCL gather).  Kept next to the package so that it travels with
a snapshot of the repository; git-ignored.

``RT_MI355_LIB`` names another build of the same ABI to load instead
(measurements that compare two builds in one process: scripts/ab_place.py).
The laboratory build of rounds 2-4 (probes, rejected kernel variants) is in
the history of this repository (last present in commit 3b63b5b).

``-ffp-contract=off`` is part of the numerical contract (see csrc/rt_math.h):
numpy never fuses a multiply into an add, and planes / spheres / conics are
held to the reference's BITS (1e-10 is only the contract's outer bound); even
aspheres run on explicitly fused arithmetic by default (1e-8 contract) and
on scipy's operations with ``exact_asphere``.
"""
import glob
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_mi355.so")
UNITS = ["rt_engine.hip", "rt_consumers.hip", "rt_comm.hip"]
SOURCES = [os.path.join(CSRC, u) for u in UNITS]
HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [
    os.path.join(HERE, "..", "include", "rt_mi355.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         "-fPIC"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librt_mi355.so")
    return exe


def _stale(lib, files):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    return any(os.path.getmtime(f) > t for f in files)


def stale():
    return _stale(LIB, SOURCES + HEADERS)


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd),
                                                      res.stdout))
    return res.stdout


def _build(lib, units, defines, tag, verbose):
    """One object per translation unit (compiled side by side), one link."""
    objdir = os.path.join(HERE, "build", tag)
    os.makedirs(objdir, exist_ok=True)
    cc = hipcc()
    objs = [os.path.join(objdir, u[:-4] + ".o") for u in units]
    cmds = [[cc] + FLAGS + defines + ["-c", os.path.join(CSRC, u), "-o", o]
            for u, o in zip(units, objs)]
    with ThreadPoolExecutor(len(cmds)) as pool:
        list(pool.map(lambda c: _run(c, verbose), cmds))
    _run([cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o",
          lib + ".tmp"] + objs + ["-ldl"], verbose)
    os.replace(lib + ".tmp", lib)
    return lib


def build(force=False, verbose=False):
    """Compile the shipped engine for gfx950; returns the path of the .so."""
    if not force and not stale():
        return LIB
    return _build(LIB, UNITS, [], "product", verbose)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
