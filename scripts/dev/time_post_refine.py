"""Dev: one object's post-refinement iterations under the torch profiler (config #4's unit of work)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams, post_refine
from scorp_amd.renderer import render as render3d

dev = torch.device('cuda:0')
raw = make_gaussians(100_000, 0, 50, extent=0.5, log_scale_mean=math.log(0.01))
obj = GaussianModel.from_raw(raw, 0, device=dev)
cams = ring_cameras(8, 1600, 1200, 9, device=dev)
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render3d(c, obj, pipe, bg)["render"].clamp(0, 1) for c in cams]
    masks = [(render3d(c, obj, pipe, bg)["render_alpha"] > 0.5).float() for c in cams]
    obj._features_dc.data.add_(0.3 * torch.randn_like(obj._features_dc))
opt = OptimizationParams()
post_refine(obj, cams, gts, masks, opt, iterations=8)
for iters in (24, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    post_refine(obj, cams, gts, masks, opt, iterations=iters)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("post_refine", iters, "iterations:", round(1e6 * dt / iters, 1), "us per iteration", flush=True)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    post_refine(obj, cams, gts, masks, opt, iterations=20)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=64))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=64))
