"""Does stacking H hypotheses x 15 views into ONE launch set buy anything over H launch sets of 15 views (config #3)?
    python scripts/dev/time_sweep_batch.py
Renders the S4 object from 15, 30, 60 stacked 800x800 views (the same 15 cameras repeated: the work per view is that of a
sweep hypothesis) and prints us per 15 views, with the per-kernel event table."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from scorp_amd import _C
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.multiview import ViewStack, render_stacked
from scorp_amd.rasterizer3d import PairPolicy
from scorp_amd.synthetic import make_gaussians, ring_cameras

dev = torch.device("cuda:0")
raw = make_gaussians(100_000, 0, 4, extent=0.8, log_scale_mean=math.log(0.01))
raw["xyz"][:, 0] *= 1.6
obj = GaussianModel.from_raw(raw, 0, device=dev)
cams = ring_cameras(15, 800, 800, 4, radius=3.0, device=dev)
bg = torch.zeros(3, device=dev)
out = {}
for H in (1, 2, 4):
    stack = ViewStack(cams * H, dev)
    PairPolicy.reset()
    n = render_stacked(obj, stack, bg)["num_pairs"]
    PairPolicy.mode, PairPolicy.reserve = "reserve", 2 * n + 4096
    for _ in range(6):
        render_stacked(obj, stack, bg)
    PairPolicy.drain()
    torch.cuda.synchronize()
    reps = 48 // H
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        render_stacked(obj, stack, bg)
    e1.record()
    PairPolicy.drain()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * H)
    _C.prof_enable(True)
    for _ in range(4):
        render_stacked(obj, stack, bg)
    PairPolicy.drain()
    torch.cuda.synchronize()
    k = _C.prof_collect()
    _C.prof_enable(False)
    out[f"H={H}"] = {"us_per_15_views": round(us, 1), "pairs": n,
                     "kernels_us_per_15_views": {nm: round(ms / c * 1e3 / H, 1) for nm, (ms, c) in k.items() if c}}
    PairPolicy.reset()
print(json.dumps(out, indent=1))
