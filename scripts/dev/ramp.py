"""Dev: per-view GPU time right after a synchronize (is there a ramp?)."""
import sys, time
sys.path.insert(0, '/root/repo')
import torch
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.rasterizer3d import PairPolicy
from scorp_amd.renderer import render
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams
from scorp_amd.train_view import train_view
dev = torch.device('cuda:0')
N, W, H, deg, seed, ncam = SCENES["S3"]
model = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev); model.active_sh_degree = deg
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
cams = ring_cameras(ncam, W, H, seed, device=dev)[:8]
with torch.no_grad():
    gts = [render(c, model, pipe, bg)["render"].clamp(0, 1) for c in cams]
params = [p for p in model.parameters()] if hasattr(model, "parameters") else [model._xyz, model._features_dc, model._features_rest, model._opacity, model._scaling, model._rotation]
PairPolicy.mode, PairPolicy.reserve = "reserve", 4_000_000
def step(i):
    train_view(cams[i % 8], model, pipe, bg, gts[i % 8], 0.2)
    for p in params: p.grad = None
for i in range(20): step(i)
PairPolicy.drain(); torch.cuda.synchronize()
for trial in range(3):
    time.sleep(0.05 * trial)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(30):
        step(i); ev[i + 1].record()
    t_host = time.perf_counter() - t0
    t1 = time.perf_counter(); PairPolicy.drain(); t_drain = time.perf_counter() - t1
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [ev[i].elapsed_time(ev[i + 1]) for i in range(30)]
    print("trial", trial, "wall ms", round(dt * 1e3, 2), "host enqueue ms", round(t_host * 1e3, 2), "drain ms", round(t_drain * 1e3, 2),
          "sum GPU ms", round(sum(per), 2), "first 6:", [round(v, 3) for v in per[:6]], "last 3:", [round(v, 3) for v in per[-3:]])
