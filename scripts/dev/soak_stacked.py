"""Stacked views (ScorpGs3dInputs.num_views) against single renders, bit for bit, over random view counts / sizes / models;
the brute-force 3-NN against a k-d tree with duplicate points; the fused Adam against torch.optim.Adam on odd sizes."""
import sys
sys.path.insert(0, '/root/repo')
import math
import numpy as np, torch
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.multiview import ViewStack, render_stacked
from scorp_amd.renderer import render
from scorp_amd.synthetic import make_gaussians, ring_cameras
dev = torch.device('cuda:0')


class Pipe:
    convert_SHs_python = False; compute_cov3D_python = False; debug = False; fused_activations = True; raw_outputs = True


rng = np.random.default_rng(3)
bad = 0
for it in range(30):
    V = int(rng.integers(2, 17)); N = int(rng.choice([1, 50, 3000, 30000])); deg = int(rng.integers(0, 4))
    W = int(rng.integers(8, 500)); H = 16 * int(rng.integers(1, 30))
    ls = math.log(float(rng.choice([0.01, 0.04, 0.15, 0.5])))
    tag = f"V={V} N={N} deg={deg} {W}x{H} scale={math.exp(ls):.2f}"
    try:
        m = GaussianModel.from_raw(make_gaussians(N, 3, 300 + it, extent=1.0, log_scale_mean=ls), 3, device=dev); m.active_sh_degree = deg
        cams = ring_cameras(V, W, H, it, radius=float(rng.uniform(1.5, 4.0)), device=dev)
        bg = torch.rand(3, device=dev)
        with torch.no_grad():
            out = render_stacked(m, ViewStack(cams, dev), bg)
            ok = True
            for v, cam in enumerate(cams):
                one = render(cam, m, Pipe(), bg)
                rows = slice(v * H, (v + 1) * H)
                ok = ok and torch.equal(out["render"][:, rows], one["render"]) and torch.equal(out["render_alpha"][rows], one["render_alpha"][0])
                ok = ok and torch.equal(out["render_depth_raw"][rows], one["render_depth_raw"][0]) and torch.equal(out["radii"][v], one["radii"])
        if not ok:
            bad += 1; print("STACK FAIL", tag)
    except Exception as e:   # noqa
        bad += 1; print("STACK EXC ", tag, type(e).__name__, str(e)[:300])

# 3-NN with duplicates and collinear points
from scipy.spatial import cKDTree
from simple_knn._C import distCUDA2
for n in (4, 5, 63, 64, 65, 255, 256, 257, 1000, 4097, 20011):
    p = rng.normal(size=(n, 3)).astype(np.float32)
    if n > 10:
        p[n // 2:n // 2 + 5] = p[0]            # six coincident points
        p[-7:, 1:] = 0                        # collinear tail
    d = distCUDA2(torch.tensor(p, device=dev)).cpu().numpy()
    k = min(4, n)
    dd, _ = cKDTree(p.astype(np.float64)).query(p.astype(np.float64), k=k)
    ref = (dd[:, 1:] ** 2).sum(1) / 3 if k == 4 else None
    if ref is not None and not np.allclose(d, ref, rtol=2e-5, atol=1e-12):
        bad += 1; print("KNN FAIL", n, float(np.abs(d - ref).max()))
torch.cuda.synchronize()
print("done, failures:", bad)
