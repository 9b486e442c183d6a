#!/usr/bin/env python3
"""Measure the BASELINE.json configurations other than the headline one on ONE MI355X (synthetic data):

  #2  train_3dgs loop, S2 (500k Gaussians, 1600x1200, SH3): full iterations incl. fused Adam   -> iterations/s
  #3  align rotation sweep, S4 (100k-Gaussian SH0 object, rotations_128.npz x 15 cameras, forward only) -> hypotheses/s
  #4  post_refine loop (4 x 100k SH0 objects merged as the reference does, masked L1+SSIM, colours only) -> iterations/s
  #5  2DGS surfel step, S6: python bench.py --scene S6

Prints one JSON line per config.  python scripts/measure_configs.py [--quick]
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np
import torch


def sync_time(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    from scorp_amd.align import render_views, rotation_sweep
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd import rasterizer3d as R
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd.train_view import train_view
    from scorp_amd.renderer import render
    from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
    from scorp_amd.train import PipelineParams, training_iteration
    from scorp_amd.transforms import gaussians_rotate
    import copy
    pipe = PipelineParams()
    bg = torch.zeros(3, device=dev)

    # ---- config #2: training iterations on S2 (render + loss + backward + densification stats + Adam) ----
    N, W, H, deg, seed, _ = SCENES["S2"]
    m = GaussianModel.from_raw(make_gaussians(N, deg, seed), deg, device=dev)
    m.active_sh_degree = deg
    opt = OptimizationParams()
    opt.random_background = True
    m.training_setup(opt)
    cams = ring_cameras(8, W, H, seed, device=dev)
    with torch.no_grad():
        gts = [render(c, m, pipe, bg)["render"].clamp(0, 1) for c in cams]
    it = {"n": 1000}   # stay clear of the densify / SH-degree schedule boundaries: steady-state iterations

    def train_it(i):
        it["n"] += 1
        if it["n"] % 1000 == 0:
            it["n"] += 1
        training_iteration(m, cams[i % 8], gts[i % 8], opt, pipe, bg, it["n"], densify=False)
    for i in range(5):
        train_it(i)
    n = 20 if args.quick else 100
    dt = sync_time(train_it, n)
    print(json.dumps({"config": "#2 train_3dgs loop, S2: 500k Gaussians 1600x1200 SH3, render+L1/SSIM+backward+FusedAdam, exact pair sizing (1 sync/iter)",
                      "iterations_per_s": round(1 / dt, 1), "ms_per_iteration": round(dt * 1e3, 3)}), flush=True)
    # same loop with the one-call view (scorp_gs3d_train_view) and a reserved pair buffer: no host sync in the iteration
    PairPolicy.reserve = int(max(R.LAST_NUM_PAIRS_LOG[-8:]) * 1.25) + 1024

    def train_it_fused(i):
        it["n"] += 1
        if it["n"] % 1000 == 0:
            it["n"] += 1
        training_iteration(m, cams[i % 8], gts[i % 8], opt, pipe, bg, it["n"], densify=False, fused_view=True)
    for i in range(5):
        train_it_fused(i)
    PairPolicy.drain()
    dt = sync_time(train_it_fused, n)
    PairPolicy.drain()
    print(json.dumps({"config": "#2 (one-call view) same loop through scorp_gs3d_train_view + FusedAdam, reserved pair buffer (no sync in the iteration)",
                      "iterations_per_s": round(1 / dt, 1), "ms_per_iteration": round(dt * 1e3, 3)}), flush=True)
    del m, gts

    # ---- config #3: 128-rotation sweep, forward-only renders ----
    rots = np.load(os.path.join(ROOT, "tests", "golden", "rotations_128.npz"))["rotations"]
    raw = make_gaussians(100_000, 0, 4, extent=0.8, log_scale_mean=math.log(0.01))
    raw["xyz"][:, 0] *= 1.6
    obj = GaussianModel.from_raw(raw, 0, device=dev)
    cams = ring_cameras(15, 800, 800, 4, radius=3.0, device=dev)
    tgt = copy.copy(obj)
    tgt._xyz, tgt._rotation, tgt._features_rest = obj._xyz.detach().clone(), obj._rotation.detach().clone(), obj._features_rest.detach().clone()
    gaussians_rotate(tgt, torch.tensor(rots[77], dtype=torch.float32, device=dev), fix_center=True)
    targets = render_views(tgt, cams, bg)
    nh = 16 if args.quick else 128
    from scorp_amd.align import SweepPlan
    plan = SweepPlan(obj, cams, targets, bg)     # sizing pass + graph capture, outside the timed sweep (as bench.py does)
    rotation_sweep(obj, rots[:2], cams, targets, bg, plan=plan)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ids, fit, best = rotation_sweep(obj, rots[:nh], cams, targets, bg, plan=plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"config": "#3 rotation sweep, S4: 100k-Gaussian SH0 object, %d hypotheses x 15 cameras 800x800, forward only" % nh,
                      "hypotheses_per_s": round(nh / dt, 2), "renders_per_s": round(nh * 15 / dt, 1), "best_id": best,
                      "planted_id": 77 if nh > 77 else None}), flush=True)

    # ---- config #3, the align loop's high-resolution renders (align_3dgs_clpe_9dof.py:157-169: cam.scale_resolution(1.5) up
    # to three times, cameras.py:139-148): forward-only renders of the object at 1600x1200 x 1.5^k.  The binning's LDS
    # histogram holds 36864 tiles per pass: 2400x1800 (16950 tiles) and 3600x2700 (38025, two passes) and 5400x4050
    # (85852, three passes) all stay on the LDS path.
    hi_cams = ring_cameras(4, 1600, 1200, 4, radius=3.0, device=dev)
    for k in range(4):
        w, h = hi_cams[0].resolution
        with torch.no_grad():
            for c in hi_cams:
                render(c, obj, pipe, bg)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_r = 12
            for i in range(n_r):
                render(hi_cams[i % 4], obj, pipe, bg)
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_r
        print(json.dumps({"config": f"#3 align-loop render, S4 object (100k Gaussians, SH0) at {w}x{h} ({((w + 15) // 16) * ((h + 15) // 16)} tiles), forward only, exact pair sizing",
                          "renders_per_s": round(1 / dt, 1), "ms_per_render": round(dt * 1e3, 3)}), flush=True)
        for c in hi_cams:
            c.scale_resolution(1.5)

    # ---- config #4: post-refinement iterations (colours only, masked loss), 4 objects merged ----
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

    def refine_it(i):
        out = render(cams[i % 8], m, pipe, bg)
        loss = fused_l1_ssim_loss(out["render"], gts[i % 8], 0.2, mask=masks[i % 8])
        loss.backward()
        with torch.no_grad():
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
    for i in range(5):
        refine_it(i)
    n = 20 if args.quick else 200
    dt = sync_time(refine_it, n)
    print(json.dumps({"config": "#4 post_refine loop: 4 x 100k SH0 objects as ONE model, 1600x1200, masked L1+SSIM, colours only + FusedAdam",
                      "iterations_per_s": round(1 / dt, 1), "ms_per_iteration": round(dt * 1e3, 3),
                      "s_per_800_iterations": round(800 * dt, 2)}), flush=True)
    PairPolicy.reserve = int(max(R.LAST_NUM_PAIRS_LOG[-8:]) * 1.25) + 1024

    def refine_it_fused(i):
        train_view(cams[i % 8], m, pipe, bg, gts[i % 8], 0.2, mask=masks[i % 8])
        with torch.no_grad():
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
    for i in range(5):
        refine_it_fused(i)
    PairPolicy.drain()
    dt = sync_time(refine_it_fused, n)
    PairPolicy.drain()
    print(json.dumps({"config": "#4 (one-call view) same loop through scorp_gs3d_train_view (frozen leaves get no gradient), reserved pair buffer",
                      "iterations_per_s": round(1 / dt, 1), "ms_per_iteration": round(dt * 1e3, 3),
                      "s_per_800_iterations": round(800 * dt, 2)}), flush=True)


if __name__ == "__main__":
    main()
