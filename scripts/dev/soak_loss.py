"""The fused L1 + SSIM loss (forward value, gradient) against the torch formulation over many odd shapes."""
import sys
sys.path.insert(0, '/root/repo')
import itertools
import numpy as np, torch
from scorp_amd.fused_loss import fused_l1_ssim_loss
from scorp_amd.loss import l1_loss, ssim_torch as ssim
dev = torch.device('cuda:0')
hs = [1, 2, 5, 10, 11, 12, 33, 34, 35, 63, 64, 65, 67, 68, 69, 100, 102, 103, 137, 1200]
ws = [1, 3, 9, 11, 31, 32, 33, 54, 63, 64, 65, 74, 127, 128, 129, 200, 1600]
bad = n = 0
rng = np.random.default_rng(5)
for H, W in itertools.product(hs, ws):
    if H * W > 400000 and not (H == 1200 and W == 1600):
        continue
    C = int(rng.choice([1, 3]))
    masked = bool(rng.integers(0, 2))
    lam = float(rng.choice([0.2, 0.5, 0.8]))
    g = torch.Generator(device=dev).manual_seed(H * 1000 + W)
    x = torch.rand((C, H, W), device=dev, generator=g)
    y = (x + 0.1 * torch.randn((C, H, W), device=dev, generator=g)).clamp(0, 1)
    mask = (torch.rand((1, H, W), device=dev, generator=g) > 0.3).float() if masked else None
    x1 = x.clone().requires_grad_(True)
    l1 = fused_l1_ssim_loss(x1, y, lam, mask); l1.backward()
    x2 = x.clone().requires_grad_(True)
    xm, ym = (x2 * mask, y * mask) if masked else (x2, y)
    ref = (1 - lam) * l1_loss(xm, ym) + lam * (1 - ssim(xm, ym)); ref.backward()
    scale = x2.grad.abs().max().item()
    dv, dg = abs(l1.item() - ref.item()), (x1.grad - x2.grad).abs().max().item() / max(scale, 1e-30)
    n += 1
    if not (dv < 5e-6 and dg < 2e-3):
        bad += 1
        print("FAIL", (C, H, W), "masked", masked, "lambda", lam, "value diff", dv, "grad rel max", dg)
torch.cuda.synchronize()
print("cases", n, "failures", bad)
