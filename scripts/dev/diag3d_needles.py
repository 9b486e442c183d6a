import sys
sys.path.insert(0, '/root/repo')
import math
import numpy as np, torch
from tests.util import make_case, image_weights
from tests.test_gs3d_gpu import hip_render
from oracle.gs_oracle import OracleRender
dev = torch.device('cuda:0')
seed = 1
rng = np.random.default_rng(seed)
base = dict(N=1500, W=200, H=152, deg=1, seed=40 + seed, log_scale=math.log(0.08))
kw, _ = make_case(**base)
s = kw["scales"].copy(); s[:, 0] = rng.uniform(0.5, 2.0, s.shape[0]); s[:, 1:] = rng.uniform(2e-4, 2e-3, (s.shape[0], 2)); kw["scales"] = s.astype(np.float32)
o32, o64 = OracleRender(np.float32, **kw), OracleRender(np.float64, **kw)
wc, wd, wa = image_weights(kw["H"], kw["W"], base["seed"])
for prec in ("split", "exact_fp32"):
    from scorp_amd.rasterizer3d import backward_precision
    with backward_precision(prec):
        out, t = hip_render(kw, dev)
    color, _, depth, alpha = out
    ((color * torch.tensor(wc, device=dev)).sum() + (depth * torch.tensor(wd, device=dev)).sum() + (alpha * torch.tensor(wa, device=dev)).sum()).backward()
    g32 = o32.backward(wc, wd, wa); g64 = o64.backward(wc.astype(np.float64), wd.astype(np.float64), wa.astype(np.float64))
    print(prec, "image: |hip-f32|", float(np.abs(color.detach().cpu().numpy() - o32.color).max()), "|f32-f64|", float(np.abs(o32.color - o64.color).max()), "|hip-f64|", float(np.abs(color.detach().cpu().numpy() - o64.color).max()))
    for name in ("means3D", "scales", "rotations", "opacities", "shs"):
        got = t[name].grad.detach().cpu().numpy().reshape(g32[name].shape).astype(np.float64)
        a, b, c = np.abs(got - g32[name]), np.abs(g32[name] - g64[name]), np.abs(got - g64[name])
        tot = np.abs(g64[name]).sum(); mx = np.abs(g64[name]).max()
        print(f"  {name:10s} rel-L1 |hip-f32| {a.sum()/tot:.2e} |f32-f64| {b.sum()/tot:.2e} |hip-f64| {c.sum()/tot:.2e}   max-norm {a.max()/mx:.2e} {b.max()/mx:.2e} {c.max()/mx:.2e}")
