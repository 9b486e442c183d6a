"""Worst offenders of the S6 full-size gradient comparison (HIP vs the 2-D oracle): python scripts/dev/diag_s6.py"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from oracle import gs_oracle
from oracle.gs_oracle import OracleRender2D
from tests.test_fullsize_gpu import _scene_kw
from tests.test_gs2d_gpu import hip_render2d
from tests.test_gs3d_gpu import perturbed
dev = torch.device("cuda:0")
kw = _scene_kw("S6"); kw["scale_modifier"] = 1.0
os.environ.setdefault("OMP_NUM_THREADS", str(len(os.sched_getaffinity(0))))
gs_oracle.set_parallel_backward(True, np.float32); gs_oracle.set_parallel_backward(True, np.float64)
o = OracleRender2D(np.float32, **kw)
rng = np.random.default_rng(6 + 99)
H, W = kw["H"], kw["W"]
wc = rng.normal(0, 1, (3, H, W)).astype(np.float32)
wa = rng.normal(0, 1, (7, H, W)).astype(np.float32); wa[5] *= 0.1
g = o.backward(wc, wa)
out, t = hip_render2d(kw, dev)
((out[0] * torch.tensor(wc, device=dev)).sum() + (out[2] * torch.tensor(wa, device=dev)).sum()).backward()
g64 = OracleRender2D(np.float64, **kw).backward(wc, wa)
gp = OracleRender2D(np.float32, **perturbed(kw, +1)).backward(wc, wa)
gm = OracleRender2D(np.float32, **perturbed(kw, -1)).backward(wc, wa)
geom = o.geom()
for nm in ("means2D", "means3D", "rotations", "scales"):
    ref = g[nm].astype(np.float64); got = t[nm].grad.detach().cpu().numpy().astype(np.float64).reshape(ref.shape)
    band = np.maximum.reduce([np.abs(g64[nm] - ref), np.abs(gp[nm].astype(np.float64) - ref), np.abs(gm[nm].astype(np.float64) - ref)])
    err = np.abs(got - ref); scale = np.abs(ref).max()
    excess = np.maximum(err - 3 * band, 0).max(1)
    rows = np.argsort(-excess)[:8]
    print(f"== {nm}: max |ref| {scale:.3e}; L1 err {err.sum() / np.abs(ref).sum():.2e}; rows with excess > 2e-3 max: {(excess > 2e-3 * scale).sum()}")
    for r in rows:
        sc = np.exp(0) * kw["scales"][r]
        T = geom["T"][r]
        print(f"  surfel {r}: excess/max {excess[r] / scale:.2e} err/max {err[r].max() / scale:.2e} band/max {band[r].max() / scale:.2e} "
              f"got {got[r]} ref {ref[r]} f64 {g64[nm][r]} | radius {o.radii[r]} depth {geom['depth'][r]:.3f} xy {geom['xy'][r]} scales {kw['scales'][r]} "
              f"normal_z {geom['nrm_o'][r][2]:.4f} opacity {geom['nrm_o'][r][3]:.3f}")
