"""Dev: config #4's one-call loop alone, for rocprofv3 --kernel-trace --stats."""
import math, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.rasterizer3d import PairPolicy
from scorp_amd.renderer import render
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams
from scorp_amd.train_view import train_view
dev = torch.device('cuda:0')
pipe = PipelineParams(); pipe.fused_activations = True
bg = torch.zeros(3, device=dev)
opt = OptimizationParams()
raws = [make_gaussians(100_000, 0, 50 + k, extent=0.5, log_scale_mean=math.log(0.01)) for k in range(4)]
for k, r in enumerate(raws):
    r["xyz"] += np.array([(k % 2) * 1.2 - 0.6, (k // 2) * 1.2 - 0.6, 0], np.float32)
merged = {kk: np.concatenate([r[kk] for r in raws]) for kk in raws[0]}
m = GaussianModel.from_raw(merged, 0, device=dev)
m.training_setup(opt)
for name in ("_opacity", "_rotation", "_scaling", "_xyz"):
    m.set_freeze(name, True)
cams = ring_cameras(8, 1600, 1200, 9, device=dev)
with torch.no_grad():
    pk = [render(c, m, pipe, bg) for c in cams]
    gts = [p["render"].clamp(0, 1) for p in pk]
    masks = [(p["render_alpha"] > 0.5).float() for p in pk]
PairPolicy.mode, PairPolicy.reserve = "reserve", 8_000_000
def it(i):
    train_view(cams[i % 8], m, pipe, bg, gts[i % 8], 0.2, mask=masks[i % 8])
    with torch.no_grad():
        m.optimizer.step()
        m.optimizer.zero_grad(set_to_none=True)
for i in range(5): it(i)
PairPolicy.drain(); torch.cuda.synchronize()
t0 = time.perf_counter()
th = 0.0
for i in range(100):
    h0 = time.perf_counter(); it(i); th += time.perf_counter() - h0
PairPolicy.drain(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 100
print("ms/iter", round(dt * 1e3, 3), "host enqueue ms/iter", round(th / 100 * 1e3, 3))
