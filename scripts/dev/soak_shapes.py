"""Odd image shapes and tiny scenes through both rasterizers against the oracles (forward + backward)."""
import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from tests.util import make_case, image_weights
from tests.test_gs3d_gpu import hip_render, oracle, compare_forward, compare_grads, oracle64_grads
from tests.test_gs2d_gpu import _parity_2d
dev = torch.device('cuda:0')
shapes = [(1, 1), (7, 5), (17, 33), (15, 16), (16, 15), (1601, 3), (3, 1201), (8, 8), (9, 9), (255, 257), (640, 1)]
bad = 0
for (W, H) in shapes:
    for N, deg, ls in ((1, 0, -1.5), (37, 3, -2.0), (900, 2, -2.5), (300, 1, -0.5)):
        case = dict(N=N, W=W, H=H, deg=deg, seed=W * 7 + H + N, log_scale=ls)
        try:
            kw, _ = make_case(**case)
            o = oracle(kw)
            out, t = hip_render(kw, dev)
            compare_forward(out, o)
            wc, wd, wa = image_weights(H, W, case["seed"])
            color, _, depth, alpha = out
            ((color * torch.tensor(wc, device=dev)).sum() + (depth * torch.tensor(wd, device=dev)).sum() + (alpha * torch.tensor(wa, device=dev)).sum()).backward()
            compare_grads(t, o.backward(wc, wd, wa), g64_fn=oracle64_grads(kw, wc, wd, wa))
        except Exception as e:   # noqa
            bad += 1
            print("3D FAIL", case, type(e).__name__, str(e)[:300])
        try:
            _parity_2d(dict(N=N, W=W, H=H, deg=deg, seed=case["seed"], log_scale=ls), dev)
        except Exception as e:   # noqa
            bad += 1
            print("2D FAIL", case, type(e).__name__, str(e)[:300])
torch.cuda.synchronize()
print("done, failures:", bad)
