"""A fuzz case in full: the gradient rows of one surfel (HIP, fp32 oracle, fp64 oracle, fp32 oracle on perturbed inputs).
DIAG_PICKS = seed:N:case:id,..."""
import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_gs3d_gpu import perturbed
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
for seed, nn, k, gid in [tuple(int(v) for v in p.split(":")) for p in os.environ["DIAG_PICKS"].split(",")]:
    case = fuzz_cases("2d", nn, seed)[k]
    kw, _ = make_case2d(**case)
    (color, radii, allmap), t = hip_render2d(kw, dev)
    c, am = color.detach().cpu().numpy(), allmap.detach().cpu().numpy()
    rng = np.random.default_rng(case["seed"] + 99)
    wc = rng.normal(0, 1, c.shape).astype(np.float32); wa = rng.normal(0, 1, am.shape).astype(np.float32); wa[5] *= 0.1
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    runs = {"f32": OracleRender2D(np.float32, **kw).backward(wc, wa), "f64": OracleRender2D(np.float64, **kw).backward(wc, wa),
            "f32+": OracleRender2D(np.float32, **perturbed(kw, +1)).backward(wc, wa), "f32-": OracleRender2D(np.float32, **perturbed(kw, -1)).backward(wc, wa)}
    mv = dict(kw); mv["means3D"] = (kw["means3D"] * np.float32(1 + 4e-6)).astype(np.float32)
    runs["f32 means+"] = OracleRender2D(np.float32, **mv).backward(wc, wa)
    print("seed", seed, "case", k, "surfel", gid)
    for nm in ("means3D", "means2D", "scales", "rotations"):
        print("  ", nm, "scale", float(np.abs(runs["f32"][nm]).max()), " hip", t[nm].grad.detach().cpu().numpy().reshape(runs["f32"][nm].shape)[gid])
        for r, g in runs.items():
            print("      ", r, np.asarray(g[nm])[gid])
