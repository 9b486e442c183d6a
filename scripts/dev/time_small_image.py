"""Forward-only render time of many Gaussians on SMALL images (few 64 x 64-pixel cells): the two-level binning against the
one-level one (SCORP_TWO_LEVEL_MIN_CELLS=<cells + 1> forces the latter for that size).  python scripts/dev/time_small_image.py"""
import json, math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from scorp_amd import _C
from scorp_amd.gaussian_model import GaussianModel
from scorp_amd.rasterizer3d import PairPolicy
from scorp_amd.renderer import render
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams

dev = torch.device("cuda:0")
out = {"two_level_min_cells": os.environ.get("SCORP_TWO_LEVEL_MIN_CELLS", "compiled default")}
for n, res in ((100_000, 256), (100_000, 384), (100_000, 512), (400_000, 256), (100_000, 800)):
    m = GaussianModel.from_raw(make_gaussians(n, 0, 4, extent=0.8, log_scale_mean=math.log(0.01)), 0, device=dev)
    cam = ring_cameras(3, res, res, 4, radius=3.0, device=dev)[1]
    bg, pipe = torch.zeros(3, device=dev), PipelineParams()
    PairPolicy.reset()
    with torch.no_grad():
        render(cam, m, pipe, bg)
        PairPolicy.mode = "reserve"
        for _ in range(5):
            render(cam, m, pipe, bg)
        PairPolicy.drain()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(40):
            render(cam, m, pipe, bg)
        e1.record()
        PairPolicy.drain()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 40
        _C.prof_enable(True)
        for _ in range(6):
            render(cam, m, pipe, bg)
        PairPolicy.drain()
        torch.cuda.synchronize()
        k = _C.prof_collect()
        _C.prof_enable(False)
    out[f"{n} Gaussians, {res}x{res} ({-(-res // 64) ** 2} cells)"] = {"us_per_render": round(us, 1),
        "kernels_us": {nm: round(ms / c * 1e3, 1) for nm, (ms, c) in k.items() if c}}
    PairPolicy.reset()
print(json.dumps(out, indent=1))
