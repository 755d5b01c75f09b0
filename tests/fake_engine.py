"""TEST DOUBLE for rayopt_amd.engine.Engine backed by the numpy oracle.

Lets the host-side glue of the drop-in (GeometricTrace, DeviceRows, packer,
dropin.accelerate) run end to end in a container without a GPU.  Lives in
tests/ only; the product never sees it.
"""
import numpy as np

from oracle import trace_numpy as tn
from oracle import consumers_numpy as cn
from rayopt_amd._lib import RT_Y, RT_U, RT_I, RT_T


class OracleEngine:
    def __init__(self):
        self.table = None
        self.rows = None
        self.keep = None
        self.w = None
        self.ms = 0.

    def upload_system(self, table):
        table = np.array(table)
        self.tables = table[None] if table.ndim == 1 else table
        self.table = self.tables[0]
        self.nsurf = self.tables.shape[1]

    def set_rays_repeat(self, y, u, copies):
        self.set_rays(np.tile(y, (copies, 1)), np.tile(u, (copies, 1)))

    def set_rays(self, y, u):
        L, n = self.nsurf, y.shape[0]
        self.nrays = n
        self.rows = {RT_Y: np.full((L, n, 3), np.nan),
                     RT_U: np.full((L, n, 3), np.nan),
                     RT_I: np.full((L, n, 3), np.nan),
                     RT_T: np.full((L, n), np.nan)}
        self.rows[RT_Y][0], self.rows[RT_U][0] = y, u
        self.rows[RT_I][0], self.rows[RT_T][0] = u, 0.
        self.valid = np.zeros(L, dtype=bool)
        self.valid[0] = True

    def upload_row(self, which, surf, src_soa):
        """rt_upload_row: (ncomp, n) SoA data into row ``surf``."""
        src = np.asarray(src_soa, dtype=float)
        self.rows[which][surf] = src if which == RT_T else src.T
        self.valid[surf] = True

    def set_weights(self, w):
        self.w = None if w is None else np.array(w)

    def set_keep_rows(self, keep):
        self.keep = None if keep is None else np.array(keep, dtype=bool)

    def set_option(self, key, value):
        pass

    def trace(self, start=1, stop=0, clip=False):
        if stop <= 0:
            stop = self.nsurf
        assert self.valid[start - 1]
        groups = len(self.tables)
        per = self.nrays//groups
        for g in range(groups):
            sl = slice(g*per, (g + 1)*per)
            Y, U, I, T = tn.propagate(
                self.tables[g], self.rows[RT_Y][start - 1][sl],
                self.rows[RT_U][start - 1][sl], start, stop, clip)
            for which, arr in ((RT_Y, Y), (RT_U, U), (RT_I, I), (RT_T, T)):
                self.rows[which][start:stop, sl] = arr
        keep = np.ones(self.nsurf, bool) if self.keep is None else self.keep
        self.valid[start:stop] = keep[start:stop]

    def trace_chunk(self, start, stop, clip, chunk, nchunks):
        """rt_trace_chunk on the double: the whole trace when the last chunk
        is asked for (rays are independent: the union of the chunks)."""
        if chunk == nchunks - 1:
            self.trace(start, stop, clip)

    def download(self, which, lo, hi, out=None):
        assert self.valid[lo:hi].all(), "row holds no data"
        a = self.rows[which][lo:hi]
        a = a if which == RT_T else np.ascontiguousarray(
            a.transpose(0, 2, 1))
        if out is not None:
            out[...] = a
            return out
        return a

    def download_xy(self, which, surf, out=None):
        assert self.valid[surf], "row holds no data"
        a = np.ascontiguousarray(self.rows[which][surf].T[:2])
        if out is not None:
            out[...] = a
            return out
        return a

    def row_stats(self, surf, group_rays, ngroups, ref=-1):
        s = cn.row_stats(self.rows[RT_Y][surf], int(group_rays), self.w,
                         None if ref is None or ref < 0 else int(ref))
        return np.c_[s, np.full(len(s), np.nan)]

    def kernel_ms(self):
        return self.ms

    def rms(self, surf, ref=-1):
        return cn.rms(self.rows[RT_Y][surf], self.w, None if ref < 0 else ref)

    def spot_stats(self, surf, group_rays, ngroups):
        assert group_rays*ngroups == self.nrays
        return cn.spot_stats(self.rows[RT_Y][surf], group_rays, self.w)

    def opd_rays(self, args):
        L = self.nsurf
        after, image = int(args["after"]), int(args["image"])
        origins = np.zeros((L, 3))
        origins[after] = args["shift"]          # only the difference is used
        frames = [None]*L
        if args["rot_after"]:
            frames[after] = np.array(args["r_after"]).reshape(3, 3)
        if args["rot_image"]:
            frames[image] = np.array(args["r_image"]).reshape(3, 3)
        n = np.zeros(L)
        n[0], n[after] = args["n0"], args["n_after"]
        return cn.opd_rays(self.rows[RT_Y], self.rows[RT_U], self.rows[RT_T],
                           n, int(args["ref"]), origins, frames,
                           bool(args["finite"]), float(args["radius"]),
                           float(args["lscale"]), after - L, image - L)

    def row_rmax(self, surf):
        return np.hypot(*self.rows[RT_Y][surf][:, :2].T).max()

    def download_ray(self, which, ray):
        a = self.rows[which][:, ray]
        return np.array(a)

    def refocus_shift(self, surf):
        w = self.w if self.w is not None else \
            np.ones(self.nrays)/self.nrays
        return cn.refocus_shift(self.rows[RT_Y][surf], self.rows[RT_I][surf],
                                w)


def _aim_pupil(self, seeds, args):
    """Test double of rt_aim_pupil: the same host+device aiming code
    (rayopt_amd/csrc/rt_aim.h) compiled for the host."""
    import ctypes
    from conftest import build_hostemu
    from rayopt_amd import _lib
    lib = ctypes.CDLL(build_hostemu())
    seeds = np.ascontiguousarray(seeds, dtype=_lib.AIM_SEED_DTYPE)
    args = np.ascontiguousarray(args, dtype=_lib.AIM_ARGS_DTYPE)
    table = np.ascontiguousarray(self.tables)
    nf = len(seeds)
    z, a = np.empty(nf), np.empty((nf, 2, 2))
    status = np.empty(nf, dtype=np.int32)
    ptr = lambda arr: ctypes.c_void_p(arr.ctypes.data)   # noqa: E731
    lib.emu_aim_pupil(ptr(table), self.nsurf, ptr(seeds), nf, ptr(args),
                      ptr(z), ptr(a), ptr(status))
    return z, a, status


def _generate_rays(self, fields, pupil_xy):
    """Test double of rt_generate_rays: the generation arithmetic compiled
    for the host (tests/hostemu)."""
    import ctypes
    from conftest import build_hostemu
    lib = ctypes.CDLL(build_hostemu())
    fields = np.ascontiguousarray(fields)
    pupil = np.ascontiguousarray(pupil_xy, dtype=float)
    n = len(fields)*len(pupil)
    Y, U = np.empty((n, 3)), np.empty((n, 3))
    s0 = np.ascontiguousarray(self.table[:1])
    lib.emu_generate(ctypes.c_void_p(fields.ctypes.data), len(fields),
                     ctypes.c_void_p(pupil.ctypes.data),
                     ctypes.c_int64(len(pupil)),
                     ctypes.c_void_p(s0.ctypes.data),
                     ctypes.c_void_p(Y.ctypes.data),
                     ctypes.c_void_p(U.ctypes.data))
    self.set_rays(Y, U)
    return n


OracleEngine.generate_rays = _generate_rays
OracleEngine.aim_pupil = _aim_pupil
