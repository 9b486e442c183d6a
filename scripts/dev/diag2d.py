import sys, math, json
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import fuzz_cases
from tests.test_gs2d_gpu import hip_render2d
from tests.test_oracle2d_cpu import make_case2d
from oracle.gs_oracle import OracleRender2D
dev = torch.device('cuda:0')
import os
SEED, NN = int(os.environ.get("DIAG_SEED", "20261004")), int(os.environ.get("DIAG_N", "32"))
for k in [int(v) for v in os.environ.get("DIAG_CASES", "16,2,7").split(",")]:
    case = fuzz_cases("2d", NN, SEED)[k]
    kw, _ = make_case2d(**case)
    o = OracleRender2D(np.float32, **kw)
    o64 = OracleRender2D(np.float64, **kw)
    out, t = hip_render2d(kw, dev)
    color, radii, allmap = out
    c, am = color.detach().cpu().numpy(), allmap.detach().cpu().numpy()
    print("case", k, case)
    for ch in range(7):
        d = np.abs(am[ch] - o.allmap[ch]); d64 = np.abs(o.allmap[ch] - o64.allmap[ch])
        print("  allmap ch", ch, "max diff hip-vs-f32", d.max(), "at", np.unravel_index(d.argmax(), d.shape), "| f32-vs-f64", d64.max())
    rng = np.random.default_rng(case["seed"] + 99)
    wc = rng.normal(0, 1, c.shape).astype(np.float32)
    wa = rng.normal(0, 1, am.shape).astype(np.float32)
    wa[5] *= 0.1
    ((color * torch.tensor(wc, device=dev)).sum() + (allmap * torch.tensor(wa, device=dev)).sum()).backward()
    g = o.backward(wc, wa); g64 = o64.backward(wc, wa)
    for nm in ("means3D", "means2D", "rotations", "scales", "opacities"):
        got = t[nm].grad.detach().cpu().numpy().reshape(g[nm].shape)
        err = np.abs(got - g[nm]).reshape(got.shape[0], -1).max(1)
        e64 = np.abs(g[nm] - g64[nm]).reshape(got.shape[0], -1).max(1)
        scale = np.abs(g[nm]).max()
        worst = np.argsort(-err)[:4]
        print("  grad", nm, "scale", scale, "worst ids", worst.tolist(), "err/scale", (err[worst] / scale).tolist(), "oracle f32-f64 at same ids", (e64[worst] / scale).tolist(),
              "radii", o.radii[worst].tolist())
    # which channel weights matter: redo backward with only one allmap channel at a time for the worst id of means3D
    got = t["means3D"].grad.detach().cpu().numpy()
    wid = int(np.argsort(-np.abs(got - g["means3D"]).max(1))[0])
    for ch in list(range(7)) + ["color"]:
        out2, t2 = hip_render2d(kw, dev)
        col2, _, am2 = out2
        w1 = np.zeros_like(wa); wcz = np.zeros_like(wc)
        if ch == "color": wcz = wc
        else: w1[ch] = wa[ch]
        ((col2 * torch.tensor(wcz, device=dev)).sum() + (am2 * torch.tensor(w1, device=dev)).sum()).backward()
        gg = o.backward(wcz, w1); gg64 = o64.backward(wcz, w1)
        a = t2["means3D"].grad.detach().cpu().numpy()[wid]
        print("    id", wid, "channel", ch, "hip", a, "f32", gg["means3D"][wid], "f64", gg64["means3D"][wid])
