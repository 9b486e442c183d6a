"""Dev: per-iteration wall time (synchronised) around the densification iterations of a fused-view run at S3 size."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams, training_iteration, _drain_reservation
from scorp_amd.renderer import render as render3d

dev = torch.device('cuda:0')
N, deg = 1_000_000, 3
model = GaussianModel.from_raw(make_gaussians(N, deg, 11), deg, device=dev); model.active_sh_degree = deg
cams = ring_cameras(8, 1600, 1200, 4, device=dev)
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render3d(c, model, pipe, bg)["render"].clamp(0, 1).clone() for c in cams]
opt = OptimizationParams()
opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval, opt.random_background = 100, 100, 3000, False
model.training_setup(opt)
for it in range(1, 421):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    training_iteration(model, cams[it % 8], gts[it % 8], opt, pipe, bg, it, scene_extent=3.0, fused_view=True)
    torch.cuda.synchronize(); dt = 1e3 * (time.perf_counter() - t0)
    if it % 100 in (99, 0, 1, 2, 3, 50) or it < 4:
        print(it, "N", model.get_xyz.shape[0], round(dt, 2), "ms", "alloc MB", round(torch.cuda.memory_allocated() / 1e6), "reserved", round(torch.cuda.memory_reserved() / 1e6), flush=True)
    if it % 32 == 0:
        _drain_reservation()
