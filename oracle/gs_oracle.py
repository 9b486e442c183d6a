"""ctypes front-end of the CPU oracle (oracle/gs3d_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product path (scorp_amd/ and the shim packages).  PARITY UNPINNED (see the C file's header).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(force=False):
    """Compile the two oracle flavours with gcc (a few seconds)."""
    targets = [os.path.join(_HERE, f"libgs_oracle_{s}.so") for s in ("f32", "f64")]
    if force or not all(os.path.exists(t) for t in targets):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return targets


def _lib(dtype):
    dtype = np.dtype(dtype)
    key = "f32" if dtype == np.float32 else "f64"
    if key not in _LIBS:
        path = os.path.join(_HERE, f"libgs_oracle_{key}.so")
        if not os.path.exists(path):
            build()
        lib = ctypes.CDLL(path)
        lib.gs3d_oracle_forward.restype = ctypes.c_void_p
        lib.gs3d_oracle_num_pairs.restype = ctypes.c_int64
        lib.gs3d_oracle_num_pairs.argtypes = [ctypes.c_void_p]
        lib.gs3d_oracle_free.argtypes = [ctypes.c_void_p]
        _LIBS[key] = lib
    return _LIBS[key]


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def set_parallel_backward(on, dtype=np.float32):
    """Tiles of the backward replay in parallel (atomic accumulation; order no longer deterministic): the CPU-baseline
    timing and the full-size parity tests.  Both rasterizers' oracles."""
    _lib(dtype).gs3d_oracle_parallel_backward(ctypes.c_int(1 if on else 0))
    _lib(dtype).gs2d_oracle_parallel_backward(ctypes.c_int(1 if on else 0))


class OracleRender:
    """One forward pass; keeps the inputs alive because the C state borrows their pointers."""

    def __init__(self, dtype, means3D, opacities, view, proj, campos, bg, W, H, tanfovx, tanfovy,
                 shs=None, sh_degree=0, colors_precomp=None, scales=None, rotations=None,
                 cov3D_precomp=None, scale_modifier=1.0):
        self.dtype = np.dtype(dtype)
        self.real = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        self.lib = _lib(self.dtype)
        c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a), dtype=self.dtype)
        self.means3D = c(means3D).reshape(-1, 3)
        self.N = self.means3D.shape[0]
        self.opacities = c(opacities).reshape(-1)
        self.shs = c(shs)
        self.colors_precomp = c(colors_precomp)
        self.scales = c(scales)
        self.rotations = c(rotations)
        self.cov3D_precomp = c(cov3D_precomp)
        assert (self.shs is None) != (self.colors_precomp is None), "exactly one of shs / colors_precomp"
        assert (self.cov3D_precomp is None) != (self.scales is None), "exactly one of scales+rotations / cov3D_precomp"
        self.K = 0 if self.shs is None else self.shs.shape[1]
        self.deg = int(sh_degree)
        self.W, self.H = int(W), int(H)
        self.view = c(view).reshape(16)
        self.proj = c(proj).reshape(16)
        self.campos = c(campos).reshape(3)
        self.bg = c(bg).reshape(3)
        self.color = np.zeros((3, self.H, self.W), self.dtype)
        self.depth = np.zeros((1, self.H, self.W), self.dtype)
        self.alpha = np.zeros((1, self.H, self.W), self.dtype)
        self.radii = np.zeros(self.N, np.int32)
        r = self.real
        self.state = self.lib.gs3d_oracle_forward(
            ctypes.c_int(self.N), ctypes.c_int(self.K), ctypes.c_int(self.deg), ctypes.c_int(self.W),
            ctypes.c_int(self.H), r(tanfovx), r(tanfovy), r(scale_modifier),
            _p(self.bg), _p(self.view), _p(self.proj), _p(self.campos),
            _p(self.means3D), _p(self.shs), _p(self.colors_precomp), _p(self.opacities),
            _p(self.scales), _p(self.rotations), _p(self.cov3D_precomp), ctypes.c_int(0),
            _p(self.color), _p(self.radii), _p(self.depth), _p(self.alpha))
        if not self.state:
            raise MemoryError("oracle forward failed")
        self.state = ctypes.c_void_p(self.state)
        self.num_pairs = int(self.lib.gs3d_oracle_num_pairs(self.state))
        self.tiles_x = (self.W + 15) // 16
        self.tiles_y = (self.H + 15) // 16

    def geom(self):
        xy = np.zeros((self.N, 2), self.dtype); depth = np.zeros(self.N, self.dtype)
        conic_o = np.zeros((self.N, 4), self.dtype); rgb = np.zeros((self.N, 3), self.dtype)
        rect = np.zeros((self.N, 4), np.int32)
        self.lib.gs3d_oracle_geom(self.state, _p(xy), _p(depth), _p(conic_o), _p(rgb), _p(rect))
        return dict(xy=xy, depth=depth, conic_o=conic_o, rgb=rgb, rect=rect)

    def tiles(self):
        tile_start = np.zeros(self.tiles_x * self.tiles_y + 1, np.int64)
        point_list = np.zeros(max(self.num_pairs, 1), np.int32)
        self.lib.gs3d_oracle_tiles(self.state, _p(tile_start), _p(point_list))
        return tile_start, point_list[: self.num_pairs]

    def backward(self, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None):
        c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a), dtype=self.dtype)
        dc, dd, da = c(dL_dcolor), c(dL_ddepth), c(dL_dalpha)
        N, K = self.N, self.K
        g = dict(means3D=np.zeros((N, 3), self.dtype), means2D=np.zeros((N, 3), self.dtype),
                 shs=np.zeros((N, K, 3), self.dtype) if K else None, colors_precomp=np.zeros((N, 3), self.dtype),
                 opacities=np.zeros((N, 1), self.dtype), scales=np.zeros((N, 3), self.dtype),
                 rotations=np.zeros((N, 4), self.dtype), cov3D_precomp=np.zeros((N, 6), self.dtype))
        self.lib.gs3d_oracle_backward(self.state, _p(dc), _p(dd), _p(da), _p(g["means3D"]), _p(g["means2D"]),
                                      _p(g["shs"]), _p(g["colors_precomp"]), _p(g["opacities"]), _p(g["scales"]),
                                      _p(g["rotations"]), _p(g["cov3D_precomp"]))
        return g

    def close(self):
        if getattr(self, "state", None):
            self.lib.gs3d_oracle_free(self.state)
            self.state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class OracleRender2D:
    """2DGS (surfel) oracle: color[3,H,W], radii[N], allmap[7,H,W]; scales are [N,2]."""

    def __init__(self, dtype, means3D, opacities, view, proj, campos, bg, W, H, tanfovx, tanfovy,
                 shs=None, sh_degree=0, colors_precomp=None, scales=None, rotations=None, transmat_precomp=None,
                 scale_modifier=1.0):
        self.dtype = np.dtype(dtype)
        self.real = ctypes.c_float if self.dtype == np.float32 else ctypes.c_double
        self.lib = _lib(self.dtype)
        self.lib.gs2d_oracle_forward.restype = ctypes.c_void_p
        self.lib.gs2d_oracle_num_pairs.restype = ctypes.c_int64
        self.lib.gs2d_oracle_num_pairs.argtypes = [ctypes.c_void_p]
        self.lib.gs2d_oracle_free.argtypes = [ctypes.c_void_p]
        c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a), dtype=self.dtype)
        self.means3D = c(means3D).reshape(-1, 3)
        self.N = self.means3D.shape[0]
        self.opacities = c(opacities).reshape(-1)
        self.shs, self.colors_precomp = c(shs), c(colors_precomp)
        self.scales, self.rotations, self.transmat_precomp = c(scales), c(rotations), c(transmat_precomp)
        assert (self.shs is None) != (self.colors_precomp is None)
        assert (self.transmat_precomp is None) != (self.scales is None)
        self.K = 0 if self.shs is None else self.shs.shape[1]
        self.W, self.H = int(W), int(H)
        self.view, self.proj = c(view).reshape(16), c(proj).reshape(16)
        self.campos, self.bg = c(campos).reshape(3), c(bg).reshape(3)
        self.color = np.zeros((3, self.H, self.W), self.dtype)
        self.allmap = np.zeros((7, self.H, self.W), self.dtype)
        self.radii = np.zeros(self.N, np.int32)
        r = self.real
        st = self.lib.gs2d_oracle_forward(
            ctypes.c_int(self.N), ctypes.c_int(self.K), ctypes.c_int(int(sh_degree)), ctypes.c_int(self.W), ctypes.c_int(self.H),
            r(tanfovx), r(tanfovy), r(scale_modifier), _p(self.bg), _p(self.view), _p(self.proj), _p(self.campos),
            _p(self.means3D), _p(self.shs), _p(self.colors_precomp), _p(self.opacities), _p(self.scales),
            _p(self.rotations), _p(self.transmat_precomp), _p(self.color), _p(self.radii), _p(self.allmap))
        if not st:
            raise MemoryError("2D oracle forward failed")
        self.state = ctypes.c_void_p(st)
        self.num_pairs = int(self.lib.gs2d_oracle_num_pairs(self.state))
        self.tiles_x, self.tiles_y = (self.W + 15) // 16, (self.H + 15) // 16

    def geom(self):
        N = self.N
        T = np.zeros((N, 9), self.dtype); xy = np.zeros((N, 2), self.dtype); depth = np.zeros(N, self.dtype)
        nrm_o = np.zeros((N, 4), self.dtype); rgb = np.zeros((N, 3), self.dtype); rect = np.zeros((N, 4), np.int32)
        self.lib.gs2d_oracle_geom(self.state, _p(T), _p(xy), _p(depth), _p(nrm_o), _p(rgb), _p(rect))
        return dict(T=T, xy=xy, depth=depth, nrm_o=nrm_o, rgb=rgb, rect=rect)

    def tiles(self):
        ts = np.zeros(self.tiles_x * self.tiles_y + 1, np.int64)
        pl = np.zeros(max(self.num_pairs, 1), np.int32)
        self.lib.gs2d_oracle_tiles(self.state, _p(ts), _p(pl))
        return ts, pl[: self.num_pairs]

    def backward(self, dL_dcolor=None, dL_dallmap=None):
        c = lambda a: None if a is None else np.ascontiguousarray(np.asarray(a), dtype=self.dtype)
        dc, da = c(dL_dcolor), c(dL_dallmap)
        N, K = self.N, self.K
        g = dict(means3D=np.zeros((N, 3), self.dtype), means2D=np.zeros((N, 3), self.dtype),
                 shs=np.zeros((N, K, 3), self.dtype) if K else None, colors_precomp=np.zeros((N, 3), self.dtype),
                 opacities=np.zeros((N, 1), self.dtype), scales=np.zeros((N, 2), self.dtype),
                 rotations=np.zeros((N, 4), self.dtype), transmat=np.zeros((N, 9), self.dtype))
        self.lib.gs2d_oracle_backward(self.state, _p(dc), _p(da), _p(g["means3D"]), _p(g["means2D"]), _p(g["shs"]),
                                      _p(g["colors_precomp"]), _p(g["opacities"]), _p(g["scales"]), _p(g["rotations"]),
                                      _p(g["transmat"]))
        return g

    def close(self):
        if getattr(self, "state", None):
            self.lib.gs2d_oracle_free(self.state)
            self.state = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
