"""Build the plain-C oracle (oracle/trace_c.c) with gcc and load it.
TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "trace_c.c")
LIB = os.path.join(HERE, "liboracle_c.so")
HDR = os.path.join(HERE, "..", "include", "rt_mi355.h")


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < max(
            os.path.getmtime(SRC), os.path.getmtime(HDR)):
        subprocess.check_call(["gcc", "-O2", "-std=c11", "-ffp-contract=off",
                               "-fopenmp", "-fPIC", "-shared", "-o", LIB, SRC,
                               "-lm"])
    return LIB


def propagate(table, y, u, start=1, stop=None, clip=False, out=None):
    """Same signature and outputs as oracle.trace_numpy.propagate; ``out``:
    (Y, U, I, T) of an earlier call to write into (pages already touched)."""
    dll = ctypes.CDLL(build())
    table = np.ascontiguousarray(table)
    idx = range(len(table))[start:stop]
    a, b = idx.start, max(idx.start, idx.stop)
    y = np.ascontiguousarray(y, dtype=float)
    u = np.ascontiguousarray(u, dtype=float)
    n = y.shape[0]
    if out is None:
        Y = np.empty((b - a, n, 3))
        U = np.empty_like(Y)
        I = np.empty_like(Y)
        T = np.empty((b - a, n))
    else:
        Y, U, I, T = out
        assert Y.shape == U.shape == I.shape == (b - a, n, 3)
        assert T.shape == (b - a, n)
    ptr = lambda arr: ctypes.c_void_p(arr.ctypes.data)   # noqa: E731
    rc = dll.oracle_propagate(ptr(table), a, b, int(bool(clip)), ptr(y),
                              ptr(u), ctypes.c_int64(n), ptr(Y), ptr(U),
                              ptr(I), ptr(T))
    assert rc == 0
    return Y, U, I, T
