"""Dev: where the reference's own call pattern (bench.py `value_dropin`) spends its 11 ms per view at S3 size:
python scripts/dev/prof_dropin.py  -> torch profiler table of device time by operator (SSIM=ssim_torch,
TORCH_ACTIVATIONS=1: the torch formulations the package replaces by itself)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.renderer import render
from scorp_amd.renderer2d import GaussianModel2D, render as render2d
from scorp_amd.rasterizer3d import PairPolicy
from scorp_amd import loss as L

dev = torch.device("cuda:0")
SURFELS = bool(int(os.environ.get("SURFELS", "0")))      # train_2dgs.py:95-150: + normal-consistency / distortion terms in torch
if SURFELS:
    model = GaussianModel2D.from_raw(make_gaussians(1_000_000, 3, 11, scale_dims=2), 3, device=dev)
    render = render2d
else:
    model = GaussianModel.from_raw(make_gaussians(1_000_000, 3, 11), 3, device=dev)
model.active_sh_degree = 3
cams = ring_cameras(8, 1600, 1200, 4, device=dev)
class pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False   # the reference's PipelineParams
if os.environ.get("TORCH_ACTIVATIONS"):
    pipe.fused_activations = False
bg = torch.zeros(3, device=dev)
PairPolicy.mode = "exact"
with torch.no_grad():
    gts = [render(c, model, pipe, bg)["render"].clamp(0, 1).clone() for c in cams]
params = [model._xyz, model._features_dc, model._features_rest, model._opacity, model._scaling, model._rotation]
ssim = getattr(L, os.environ.get("SSIM", "ssim"))


def step(i):
    out = render(cams[i % 8], model, pipe, bg)
    img = out["render"]
    loss = 0.8 * L.l1_loss(img, gts[i % 8]) + 0.2 * (1.0 - ssim(img, gts[i % 8]))
    if SURFELS:
        normal_error = (1 - (out["render_normal"] * out["surf_normal"]).sum(dim=0))[None]
        loss = loss + 0.05 * normal_error.mean() + 100.0 * out["render_dist"].mean()
    loss.backward()
    for p in params:
        p.grad = None


for i in range(4):
    step(i)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(16):
    step(i)
torch.cuda.synchronize(); print("ms per view", round((time.perf_counter() - t0) / 16 * 1e3, 3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for i in range(8):
        step(i)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=28, max_name_column_width=60))
