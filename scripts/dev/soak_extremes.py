"""Extreme splat shapes / opacities through both rasterizers against the oracles (forward + backward)."""
import sys
sys.path.insert(0, '/root/repo')
import math
import numpy as np, torch
from tests.util import make_case, image_weights
from tests.test_gs3d_gpu import hip_render, oracle, compare_forward, compare_grads, oracle64_grads
from tests.test_gs2d_gpu import _parity_2d
from tests.test_oracle2d_cpu import make_case2d
dev = torch.device('cuda:0')
bad = 0


def run3d(tag, kw, seed):
    global bad
    try:
        o = oracle(kw)
        out, t = hip_render(kw, dev)
        compare_forward(out, o)
        wc, wd, wa = image_weights(kw["H"], kw["W"], seed)
        color, _, depth, alpha = out
        ((color * torch.tensor(wc, device=dev)).sum() + (depth * torch.tensor(wd, device=dev)).sum() + (alpha * torch.tensor(wa, device=dev)).sum()).backward()
        compare_grads(t, o.backward(wc, wd, wa), g64_fn=oracle64_grads(kw, wc, wd, wa))
        print("3D ok  ", tag, "max alpha-image diff", float(np.abs(out[3].detach().cpu().numpy() - o.alpha).max()))
    except Exception as e:   # noqa
        bad += 1
        print("3D FAIL", tag, type(e).__name__, str(e)[:400])


def run2d(tag, case, kw):
    global bad
    try:
        _parity_2d(case, dev, kw=kw)
        print("2D ok  ", tag)
    except Exception as e:   # noqa
        bad += 1
        print("2D FAIL", tag, type(e).__name__, str(e)[:400])


for seed in (1, 2, 3):
    rng = np.random.default_rng(seed)
    base = dict(N=1500, W=200, H=152, deg=1, seed=40 + seed, log_scale=math.log(0.08))
    # needles: one long axis, two (3-D) / one (2-D) tiny
    kw, _ = make_case(**base)
    s = kw["scales"].copy(); s[:, 0] = rng.uniform(0.5, 2.0, s.shape[0]); s[:, 1:] = rng.uniform(2e-4, 2e-3, (s.shape[0], 2)); kw["scales"] = s.astype(np.float32)
    run3d(f"needles seed {seed}", kw, base["seed"])
    # pancakes seen from anywhere: two long axes, one tiny
    kw, _ = make_case(**base)
    s = kw["scales"].copy(); s[:, :2] = rng.uniform(0.3, 1.0, (s.shape[0], 2)); s[:, 2] = 1e-4; kw["scales"] = s.astype(np.float32)
    run3d(f"pancakes seed {seed}", kw, base["seed"])
    # opacities at the two ends
    kw, _ = make_case(**base)
    o = kw["opacities"].copy(); o[::2] = rng.uniform(0.0035, 0.0045, o[::2].shape); o[1::2] = rng.uniform(0.985, 0.99999, o[1::2].shape); kw["opacities"] = o.astype(np.float32)
    run3d(f"opacity ends seed {seed}", kw, base["seed"])
    # 2-D: long thin surfels, large flat ones, opacity ends
    kw2, _ = make_case2d(**base)
    s = kw2["scales"].copy(); s[:, 0] = rng.uniform(0.5, 2.0, s.shape[0]); s[:, 1] = rng.uniform(2e-4, 2e-3, s.shape[0]); kw2["scales"] = s.astype(np.float32)
    run2d(f"needles seed {seed}", base, kw2)
    kw2, _ = make_case2d(**base)
    kw2["scales"] = rng.uniform(0.3, 1.2, kw2["scales"].shape).astype(np.float32)
    run2d(f"large surfels seed {seed}", base, kw2)
    kw2, _ = make_case2d(**base)
    o = kw2["opacities"].copy(); o[::2] = rng.uniform(0.0035, 0.0045, o[::2].shape); o[1::2] = rng.uniform(0.985, 0.99999, o[1::2].shape); kw2["opacities"] = o.astype(np.float32)
    run2d(f"opacity ends seed {seed}", base, kw2)
torch.cuda.synchronize()
print("done, failures:", bad)
