"""Dev: where a full training iteration at S3 size spends its time (one-call view + FusedAdam + statistics + densification
every 100): python scripts/dev/time_train_iter.py [iterations]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams, train
from scorp_amd.renderer import render as render3d
from scorp_amd.renderer2d import GaussianModel2D, render as render2d

dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
N, deg = 1_000_000, 3
SURFELS = bool(int(os.environ.get("SURFELS", "0")))
raw = make_gaussians(N, deg, 11)
if SURFELS:
    raw["scaling"] = raw["scaling"][:, :2].copy()
model = (GaussianModel2D if SURFELS else GaussianModel).from_raw(raw, deg, device=dev); model.active_sh_degree = deg
render3d = render2d if SURFELS else render3d
KW = dict(surfels=True) if SURFELS else {}
if os.environ.get("NO_FUSED_STEP"):      # A/B: the optimizer step as a separate launch (round 5's iteration)
    KW["fused_step"] = False
cams = ring_cameras(8, 1600, 1200, 4, device=dev)
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render3d(c, model, pipe, bg)["render"].clamp(0, 1).clone() for c in cams]
for dens in (False, True):
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval = (100, 100, 3000) if dens else (10 ** 9, 10 ** 9, 10 ** 9)
    opt.random_background = False
    model.optimizer = None
    train(model, cams, gts, opt, pipe, iterations=40, scene_extent=3.0, fused_view=True, **KW)   # warm-up, sizes the reservation
    torch.cuda.synchronize(); t0 = time.perf_counter()
    train(model, cams, gts, opt, pipe, iterations=iters, scene_extent=3.0, fused_view=True, **KW)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("densify" if dens else "no densify", "N", model.get_xyz.shape[0], "it/s", round(iters / dt, 1), "ms/it", round(1e3 * dt / iters, 3), flush=True)
if os.environ.get("PROFILE"):
    from torch.profiler import profile, ProfilerActivity
    opt = OptimizationParams(); opt.densify_from_iter = 10 ** 9; opt.random_background = False
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        train(model, cams, gts, opt, pipe, iterations=20, scene_extent=3.0, fused_view=True, **KW)
        torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
    print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=15, max_name_column_width=60))
