"""Build librt_mi355.so (HIP, gfx950) in-tree with hipcc.

The shared library is kept next to the package (``rayopt_amd/librt_mi355.so``)
so that it travels with a snapshot of the repository; it is git-ignored.
``-ffp-contract=off`` is part of the numerical contract (see csrc/rt_math.h):
numpy never fuses a multiply into an add and parity with the reference is
judged at 1e-10.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librt_mi355.so")
SOURCES = [os.path.join(CSRC, "rt_engine.hip")]
HEADERS = [os.path.join(CSRC, "rt_math.h"), os.path.join(CSRC, "rt_kernels.h"),
           os.path.join(CSRC, "rt_aim.h"),
           os.path.join(HERE, "..", "include", "rt_mi355.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
         "-fPIC", "-shared"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build librt_mi355.so")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    """Compile the HIP engine for gfx950; returns the path of the .so."""
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB + ".tmp"] + SOURCES + ["-ldl"]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                         text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stdout)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
