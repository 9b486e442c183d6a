"""Dev: host-side cost of a fused training iteration (cProfile) on a small model, where the GPU is not the limit."""
import cProfile, math, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams, train
from scorp_amd.renderer import render as render3d

dev = torch.device('cuda:0')
N, deg = int(os.environ.get("N", 50_000)), 3
model = GaussianModel.from_raw(make_gaussians(N, deg, 11), deg, device=dev); model.active_sh_degree = deg
cams = ring_cameras(8, 400, 300, 4, device=dev)
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render3d(c, model, pipe, bg)["render"].clamp(0, 1).clone() for c in cams]
opt = OptimizationParams(); opt.densify_from_iter = 10 ** 9; opt.random_background = False
train(model, cams, gts, opt, pipe, iterations=50, scene_extent=3.0, fused_view=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
train(model, cams, gts, opt, pipe, iterations=400, scene_extent=3.0, fused_view=True)
torch.cuda.synchronize(); print("us per iteration:", round(1e6 * (time.perf_counter() - t0) / 400, 1), flush=True)
pr = cProfile.Profile(); pr.enable()
train(model, cams, gts, opt, pipe, iterations=400, scene_extent=3.0, fused_view=True)
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
