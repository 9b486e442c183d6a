"""Dev: what one densify_and_prune costs at S3 size (torch profiler): python scripts/dev/time_densify.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians

dev = torch.device('cuda:0')
N, deg = int(os.environ.get("N", 1_000_000)), 3
raw = make_gaussians(N, deg, 11)
opt = OptimizationParams()
def fresh():
    m = GaussianModel.from_raw(raw, deg, device=dev); m.active_sh_degree = deg
    m.training_setup(opt)
    g = torch.Generator(device="cpu").manual_seed(1)
    m.xyz_gradient_accum += (torch.rand(N, 1, generator=g) * 6e-4).to(dev)   # ~1/3 above the 2e-4 threshold
    m.denom += 1
    m.max_radii2D += (torch.rand(N, generator=g) * 30).to(dev)
    return m
for rep in range(3):
    m = fresh()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.densify_and_prune(opt.densify_grad_threshold, 0.005, 3.0, 20)
    torch.cuda.synchronize(); print("densify_and_prune", N, "->", m.get_xyz.shape[0], round(1e3 * (time.perf_counter() - t0), 2), "ms", flush=True)
from torch.profiler import profile, ProfilerActivity
m = fresh()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    m.densify_and_prune(opt.densify_grad_threshold, 0.005, 3.0, 20)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=12, max_name_column_width=70))
