"""Dev: the fused loss forward + backward at 1600x1200 without a mask, under a dense random mask and under an object's
silhouette (15 % of the image): python scripts/dev/time_masked_loss.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd import _C
from scorp_amd.fused_loss import fused_l1_ssim_loss
dev = torch.device('cuda:0')
x = torch.rand(3, 1200, 1600, device=dev); y = torch.rand(3, 1200, 1600, device=dev)
dense = (torch.rand(1, 1200, 1600, device=dev) > 0.3).float()
obj = torch.zeros(1, 1200, 1600, device=dev); obj[:, 400:800, 500:1100] = 1.0
for name, m in (("no mask", None), ("dense random mask", dense), ("object mask, 12.5 % of the image", obj)):
    xs = x.clone().requires_grad_(True)
    for _ in range(5):
        fused_l1_ssim_loss(xs, y, 0.2, m).backward()
    _C.prof_enable(True)
    for _ in range(50):
        fused_l1_ssim_loss(xs, y, 0.2, m).backward()
    torch.cuda.synchronize()
    t = _C.prof_collect()
    _C.prof_enable(False)
    print(name, {k: round(1e3 * v[0] / max(v[1], 1), 1) for k, v in t.items() if "ssim" in k}, flush=True)
