#!/usr/bin/env python3
"""bench.py — fwd+bwd views/s of the 3DGS hot path on synthetic Gaussians (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scene S3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One step = one view of the workload: render() through the HIP rasterizer (GaussianModel activations + SH/cov in
kernel) -> 0.8*L1 + 0.2*(1-SSIM) against a resident ground-truth image -> backward to all 59 per-Gaussian
parameters (+ the means2D gradient).  The optimizer step is NOT part of the metric ("fwd+bwd views/s") and is not
in the timed region.  Inputs (parameters, cameras, GT images) are resident in HBM before the timed region.
With N > 1 every rank holds a replica of the scene (broadcast once from rank 0 over RCCL) and renders its own
views (view i -> rank i mod N): weak scaling, no collective on the data path; value = views of all ranks / time.
Rank 0 prints ONE JSON line: numbers only (what every field means, and the prose that used to ride in the line, is
DESIGN.md section 5), short enough for a log tail, with `value`, `value_exact_fp32`, `value_deterministic_backward`,
`roofline` and `cpu_baseline` as its LAST keys.
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def _sig(x, digits=3):
    """A float at `digits` significant digits (the JSON line carries numbers, not their noise)."""
    x = float(x)
    return x if x == 0 or not math.isfinite(x) else float(f"{x:.{digits}g}")


def _grad_errors(got, ref):
    got, ref = np.asarray(got, np.float64).reshape(np.asarray(ref).shape), np.asarray(ref, np.float64)
    return (float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)), float(np.abs(got - ref).sum() / max(np.abs(ref).sum(), 1e-300)))


GRAD_NAMES = ("means3D", "means2D", "opacities", "shs", "scales", "rotations")


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = True   # raw GaussianModel leaves go straight to the kernels (same numbers, no torch.cat / activations)


def kernel_algorithmic_bytes(name, N, Nvis, K, HW, D):
    """Compulsory HBM bytes of one launch of each kernel: every datum moved once."""
    per_g_in = 12 + 4 + 12 + 16                      # xyz, opacity, scale, quaternion
    return {
        "preprocess": N * (per_g_in + 16 + 4) + Nvis * (K * 12 + 48),
        "count_tiles": N * 24,                       # the 16-byte bin record + the 8-byte tile mask
        "scan_tiles": 0,
        "scatter_pairs": N * 24 + D * (8 + 16),      # two-level binning: cell keys written, then read and written again as tile buckets
        "sort_tiles": D * (8 + 4),
        "blend_forward": 4 * D * 4 + Nvis * 48 + HW * (20 + 8),      # each of a tile's 4 waves walks the tile's list
        "blend_backward": 4 * D * 4 + Nvis * 48 + HW * (20 + 8) + Nvis * 40,
        "preprocess_backward": N * (per_g_in + 16) + Nvis * (K * 12 + 48) + N * 248,
        "ssim_l1_forward": HW * 3 * (8 + 12),       # read img+gt, write 3 derivative maps
        "ssim_l1_backward": HW * 3 * (12 + 8 + 4),  # read 3 maps + img+gt, write grad
        "preprocess_2d": N * (40 + 16 + 4 + 8) + Nvis * (K * 12 + 96),
        "blend_forward_2d": D * 4 + Nvis * 96 + HW * (40 + 20),
        "blend_backward_2d": D * 4 + Nvis * 96 + HW * (40 + 20) + Nvis * 72,
        "preprocess_backward_2d": N * (40 + 16) + Nvis * (K * 12 + 96 + 80) + N * 244,
    }.get(name, 0)


def cpu_baseline(raw, cam, deg, W, H):
    """The CPU oracle (OpenMP C restatement) on ONE view of the same workload, forward + backward."""
    cores = len(os.sched_getaffinity(0))
    os.environ["OMP_NUM_THREADS"] = str(cores)     # read by libgomp when the oracle library is first loaded
    from oracle import gs_oracle
    from oracle.gs_oracle import OracleRender
    from scorp_amd.synthetic import activate
    gs_oracle.build()
    act = activate(raw)
    kw = dict(means3D=act["means3D"], opacities=act["opacities"], shs=act["shs"], sh_degree=deg, scales=act["scales"],
              rotations=act["rotations"], W=W, H=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
              view=cam.world_view_transform.cpu().numpy(), proj=cam.full_proj_transform.cpu().numpy(),
              campos=cam.camera_center.cpu().numpy(), bg=np.zeros(3, np.float32))
    gs_oracle.set_parallel_backward(True)
    w = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    t0 = time.perf_counter()
    o = OracleRender(np.float32, **kw)
    o.full_size_case = (kw, w, o.backward(w, None, None))   # kept for the full-size parity figures
    dt = time.perf_counter() - t0
    gs_oracle.set_parallel_backward(False)
    # the loss half of the step on the same host: the torch formulation of the reference's l1_loss / ssim
    # (scorp_amd/loss.py restates gs3dgs/utils/loss_utils.py:17-73), forward + backward on the oracle's image - at
    # min(cores, 32) threads and at every core (a depthwise 11x11 conv2d over 3 x 1200 x 1600 does not scale to 256
    # threads: oversubscribed it is several times slower); the baseline takes the faster of the two
    from scorp_amd.loss import l1_loss, ssim_torch as ssim
    gt = (torch.tensor(o.color) + 0.05).clamp(0, 1)
    loss_t = {}
    for nt in sorted({min(cores, 32), cores}):
        torch.set_num_threads(nt)
        img = torch.tensor(o.color).requires_grad_(True)
        (0.8 * l1_loss(img, gt) + 0.2 * (1.0 - ssim(img, gt))).backward()      # first call at this thread count: untimed
        img.grad = None
        t1 = time.perf_counter()
        (0.8 * l1_loss(img, gt) + 0.2 * (1.0 - ssim(img, gt))).backward()
        loss_t[nt] = time.perf_counter() - t1
    best_nt = min(loss_t, key=loss_t.get)
    dt_loss = loss_t[best_nt]
    return dict(value=round(1.0 / (dt + dt_loss), 5), unit="views/s", cores=cores, kind="port",
                sample="1 S3 view: OpenMP oracle render fwd+bwd + torch-CPU L1/SSIM fwd+bwd",
                render_s=round(dt, 3), loss_s=round(dt_loss, 3), loss_threads=best_nt,
                loss_s_all_cores=round(loss_t[cores], 3)), o


def small_parity(dev):
    """Quality half of the metric: PSNR / L1 of the HIP render vs the CPU oracle on BASELINE config #1 (S1)."""
    from oracle.gs_oracle import OracleRender
    from scorp_amd.synthetic import activate, scene
    from scorp_amd.refcall import render3d_reference_call as hip_render
    raw, cams, deg = scene("S1")
    act = activate(raw)
    cam = cams[0]
    kw = dict(means3D=act["means3D"], opacities=act["opacities"], shs=act["shs"], sh_degree=deg, scales=act["scales"],
              rotations=act["rotations"], W=256, H=256, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
              view=cam.world_view_transform.numpy(), proj=cam.full_proj_transform.numpy(),
              campos=cam.camera_center.numpy(), bg=np.zeros(3, np.float32))
    o = OracleRender(np.float32, **kw)
    with torch.no_grad():
        (color, _, _, _), _ = hip_render(kw, dev, requires_grad=False)
    c = color.cpu().numpy()
    mse = float(((c - o.color) ** 2).mean())
    return dict(S1_l1=_sig(np.abs(c - o.color).mean()), S1_psnr_db=(99.0 if mse == 0 else round(float(10 * math.log10(1.0 / mse)), 1)))


def s6_full_size_parity(dev):
    """View 0 of S6, HIP against the 2-D CPU oracle (OpenMP, backward with tiles in parallel): image / allmap L1 and, per
    gradient tensor of the photometric upstream gradient 1 / (3 H W), the relative L1 distances HIP - oracle32,
    HIP - oracle64 and oracle32 - oracle64: the float64 build is the third party that says which fp32 form is closer."""
    from oracle import gs_oracle
    from oracle.gs_oracle import OracleRender2D
    from scorp_amd.refcall import render2d_reference_call as hip_render2d
    from scorp_amd.synthetic import SCENES, activate, scene
    raw, cams, deg = scene("S6")
    N, W, H = SCENES["S6"][:3]
    act, cam = activate(raw), cams[0]
    kw = dict(means3D=act["means3D"], opacities=act["opacities"], shs=act["shs"], sh_degree=deg, scales=act["scales"],
              rotations=act["rotations"], W=W, H=H, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
              view=cam.world_view_transform.numpy().astype(np.float32), proj=cam.full_proj_transform.numpy().astype(np.float32),
              campos=cam.camera_center.numpy().astype(np.float32), bg=np.zeros(3, np.float32), scale_modifier=1.0)
    w = np.full((3, H, W), 1.0 / (3 * H * W), np.float32)
    gs_oracle.set_parallel_backward(True, np.float32)
    gs_oracle.set_parallel_backward(True, np.float64)
    try:
        o = OracleRender2D(np.float32, **kw)
        g = o.backward(w, None)
        g64 = OracleRender2D(np.float64, **kw).backward(w, None)
    finally:
        gs_oracle.set_parallel_backward(False, np.float32)
        gs_oracle.set_parallel_backward(False, np.float64)
    from scorp_amd.rasterizer3d import backward_precision
    rec = None
    for form in ("exact_fp32", "split"):     # both forms of the backward's pixel -> surfel reduction (scorp_gs2d_backward_ex)
        with backward_precision(form):
            out, t = hip_render2d(kw, dev)
        (out[0] * torch.tensor(w, device=dev)).sum().backward()
        if rec is None:
            c, am = out[0].detach().cpu().numpy(), out[2].detach().cpu().numpy()
            mse = float(((c - o.color) ** 2).mean())
            rec = dict(l1=_sig(np.abs(c - o.color).mean()), psnr_db=(99.0 if mse == 0 else round(float(10 * math.log10(1.0 / mse)), 1)),
                       allmap_l1_max=_sig(max(np.abs(am[ch] - o.allmap[ch]).mean() / max(np.abs(o.allmap[ch]).max(), 1.0) for ch in range(7))),
                       grad_rel_l1_o32_vs_f64={nm: _sig(_grad_errors(g[nm], g64[nm])[1]) for nm in GRAD_NAMES})
        rec["grad_rel_l1_hip_vs_o32_" + form] = {nm: _sig(_grad_errors(t[nm].grad.detach().cpu().numpy(), g[nm])[1]) for nm in GRAD_NAMES}
        rec["grad_rel_l1_hip_vs_f64_" + form] = {nm: _sig(_grad_errors(t[nm].grad.detach().cpu().numpy(), g64[nm])[1]) for nm in GRAD_NAMES}
    return rec


def _event_region(fn_lead_in, fn_timed):
    """`fn_lead_in()` (untimed, keeps the chip off its idle ramp), then `fn_timed()` between two hipEvents on the current
    stream with no host synchronisation in between.  Returns (seconds between the events, fn_timed's result)."""
    fn_lead_in()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    res = fn_timed()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e-3, res


def _max_over_ranks(dt, cdev, world):
    if world > 1:
        tt = torch.tensor([dt], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    return dt


def secondary_s6(dev, steps=40, warmup=8, cams=4, lead_in=24, parity=False):
    """Secondary record of the default run: BASELINE config #5, the 2DGS surfel step on S6 (1 M surfels, 1600x1200, SH3):
    render + 0.8 L1 + 0.2 (1 - SSIM) + normal-consistency / distortion regularisers + backward, one library call per
    view.  Same timing protocol as the headline (events behind lead-in + warm-up views)."""
    from scorp_amd import _C
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd import rasterizer3d as R
    from scorp_amd.renderer2d import GaussianModel2D, render as render2d
    from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
    from scorp_amd.train_view import train_view2d
    N, W, H, deg, seed, ncam_total = SCENES["S6"]
    model = GaussianModel2D.from_raw(make_gaussians(N, deg, seed, scale_dims=2), deg, device=dev)
    model.active_sh_degree = deg
    params = [model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity]
    my_cams = ring_cameras(ncam_total, W, H, seed, device=dev)[:cams]
    bg, pipe = torch.zeros(3, device=dev), Pipe()
    g = torch.Generator(device=dev).manual_seed(4321)
    gts = []
    with torch.no_grad():
        for cam in my_cams:
            img = render2d(cam, model, pipe, bg)["render"]
            gts.append((img + 0.05 * torch.randn(img.shape, device=dev, generator=g)).clamp(0, 1))
    Ds = list(R.LAST_NUM_PAIRS_LOG[-len(my_cams):])
    with torch.no_grad():
        nvis = float(np.mean([int((render2d(c_, model, pipe, bg)["radii"] > 0).sum()) for c_ in my_cams[:2]]))

    def step(i):   # one library call per view (scorp_gs2d_train_view): render + L1/SSIM + regularisers + backward
        train_view2d(my_cams[i % cams], model, pipe, bg, gts[i % cams], 0.2, 0.05, 100.0)
        for p in params:
            p.grad = None

    PairPolicy.mode, PairPolicy.reserve = "reserve", int(max(Ds) * 1.25) + 1024
    _C.prof_enable(True)
    for i in range(warmup):
        step(i)
    PairPolicy.drain()
    torch.cuda.synchronize()
    kern = _C.prof_collect()
    _C.prof_enable(False)
    # `value`, like the headline's, is measured with every operand of the backward at fp32 (SCORP_BACKWARD_EXACT_FP32: fp32 values,
    # fp32 MFMAs); the library's default - two fp16 terms per value under a per-hit power of two - is `value_split22`
    prev_flags = getattr(R._tls, "backward_flags", 0)
    try:
        R._tls.backward_flags = _C.BACKWARD_EXACT_FP32
        _C.prof_enable(True, only=["blend_backward_2d"])
        for i in range(4):
            step(i)
        PairPolicy.drain()
        torch.cuda.synchronize()
        kern_exact = _C.prof_collect()
        _C.prof_enable(False)
        dt_exact, _ = _event_region(lambda: [step(i) for i in range(lead_in + warmup)], lambda: [step(warmup + i) for i in range(steps)])
        R._tls.backward_flags = 0
        dt, _ = _event_region(lambda: [step(i) for i in range(lead_in + warmup)], lambda: [step(warmup + i) for i in range(steps)])
    finally:
        R._tls.backward_flags = prev_flags
    PairPolicy.drain()
    # extra: FULL 2DGS training iterations (train_2dgs.py:95-199 without the densification itself): the one-call view with the
    # optimizer step and the statistics inside it (ScorpGs2dTrainView.adam), the library's default backward
    train_its = None
    try:
        from scorp_amd.gaussian_model import OptimizationParams2D
        from scorp_amd.train import training_iteration
        opt_ = OptimizationParams2D()
        opt_.densify_from_iter, opt_.opacity_reset_interval, opt_.random_background = 1 << 30, 1 << 30, False
        opt_.depth_from_iter = 1 << 30      # (no depth / isotropic terms: the plain photometric loss + the two regularisers)
        model.optimizer = None
        model.training_setup(opt_)
        n_it = 40

        def it_(i):
            training_iteration(model, my_cams[i % cams], gts[i % cams], opt_, pipe, bg, 8000 + i, scene_extent=3.0, fused_view=True, surfels=True)
        dt_it, _ = _event_region(lambda: [it_(i) for i in range(10)], lambda: [it_(10 + i) for i in range(n_it)])
        PairPolicy.drain()
        train_its = round(n_it / dt_it, 1)
    except Exception as e:   # noqa: BLE001   (an extra: never costs the record)
        train_its = f"{type(e).__name__}: {e}"[:200]
    finally:
        model.optimizer = None
        for p in params:
            p.grad = None
    PairPolicy.reset()
    kus = {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in kern.items() if cnt}
    bwd_exact_us = round(kern_exact["blend_backward_2d"][0] / max(kern_exact["blend_backward_2d"][1], 1) * 1e3, 1)
    dom = max(kus, key=kus.get)
    K = (deg + 1) ** 2
    alg = kernel_algorithmic_bytes(dom, N, nvis, K, W * H, float(np.mean(Ds)))
    traffic = None      # PMC bytes per launch of the dominant kernel, if profiles/traffic.json was collected on the loaded sources
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("source_sha") == _C.lib().scorp_source_sha().decode():
            traffic = tj.get("S6", {}).get(dom)
    dom_us = bwd_exact_us if dom == "blend_backward_2d" else kus[dom]     # the roofline is the headline form's
    rec = {"metric": "fwd+bwd views/sec (S6: 1M surfels 1600x1200 SH3, config #5)", "value": round(steps / dt_exact, 2), "unit": "views/s",
           "precision": "f32 throughout (blend backward's pixel->surfel reduction on fp32 MFMAs, SCORP_BACKWARD_EXACT_FP32)",
           "value_exact_fp32": round(steps / dt_exact, 2), "ms_per_step_exact_fp32": round(dt_exact / steps * 1e3, 4),
           "value_split22": round(steps / dt, 2), "ms_per_step_split22": round(dt / steps * 1e3, 4),
           "library_default_backward": "2-term fp16 split under a per-hit power of two (value_split22)",
           "train_iterations_per_s": train_its,
           "steps": steps, "ms_per_step": round(dt_exact / steps * 1e3, 4), "pairs_per_view_D": round(float(np.mean(Ds))), "visible": round(nvis),
           "kernels_us": kus, "kernels_us_note": "probe views of the split form; blend_backward_2d in the all-fp32 form: blend_backward_2d_exact_fp32_us",
           "blend_backward_2d_exact_fp32_us": bwd_exact_us,
           "roofline": {"bound": "hbm", "limited_by": "vector instruction issue (valu)", "kernel": dom, "achieved": round(alg / (dom_us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(alg / (dom_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_us": dom_us, "algorithmic_bytes": int(alg)}}
    if parity:
        rec["parity_full_size"] = s6_full_size_parity(dev)
    return rec


def guarded_record(prepare, run, cdev):
    """A secondary record that contains collectives, made safe for N > 1: `prepare()` is the rank-local part (it may
    raise), then ALL ranks agree that everyone got through it (one 4-byte all-reduce) before any of them enters `run(ctx)`
    - the part with the exchange steps, whose product functions agree again in front of each exchange.  A failure
    anywhere becomes {"error": ...} in the record on every rank and never leaves a rank waiting in a collective the
    others skipped; it never costs the headline line."""
    from scorp_amd.parallel import all_ok
    ctx, err = None, None
    try:
        ctx = prepare()
    except Exception as e:   # noqa: BLE001
        err = e
    if not all_ok(err is None, cdev):
        return {"error": f"{type(err).__name__}: {err}" if err is not None else "another rank failed in the local part"}
    try:
        return run(ctx)
    except Exception as e:   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}


def secondary_sweep(dev, cdev, rank, world):
    """Secondary record, every N: BASELINE config #3 as north_star scores it - the 128-rotation alignment sweep on S4 (a
    100k-Gaussian SH0 object, rotations_128.npz x 15 cameras at 800x800, forward-only renders).  Hypothesis j -> rank
    j mod N, the object is broadcast once (one flat buffer), ONE fixed-size all-gather of (id, fitness) at the end: STRONG
    scaling (128 hypotheses whatever N).  The plan (targets in the stacked layout, pair-buffer sizing pass) is built
    outside the timed region, like the model load; the timed region is score-all-my-hypotheses + gather between two
    hipEvents behind a lead-in sweep of 32 hypotheses per rank, max over ranks."""
    import copy
    from scorp_amd.align import SweepPlan, render_views, rotation_sweep
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.parallel import all_ok, broadcast_tensors
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    from scorp_amd.transforms import gaussians_rotate
    from scorp_amd import rasterizer3d as R_
    rots = np.load(os.path.join(ROOT, "tests", "golden", "rotations_128.npz"))["rotations"]
    n_obj, planted = 100_000, 77

    def prepare():
        shapes = dict(xyz=(n_obj, 3), scaling=(n_obj, 3), rotation=(n_obj, 4), opacity=(n_obj, 1), features_dc=(n_obj, 1, 3),
                      features_rest=(n_obj, 0, 3))
        t, gen_err = None, None
        try:      # rank 0 draws the object: rank-local, may raise - agreed on BEFORE anyone enters the broadcast
            t = {k: torch.empty(shp, dtype=torch.float32, device=cdev) for k, shp in shapes.items()}
            if rank == 0:
                raw = make_gaussians(n_obj, 0, 4, extent=0.8, log_scale_mean=math.log(0.01))
                raw["xyz"][:, 0] *= 1.6
                t = {k: torch.tensor(raw[k], device=cdev).reshape(shapes[k]) for k in shapes}
        except Exception as e:   # noqa: BLE001
            gen_err = e
        if not all_ok(gen_err is None, cdev):
            raise RuntimeError(f"object generation failed on some rank: {gen_err}")
        broadcast_tensors(t, src=0)
        obj = GaussianModel.from_raw({k: v.cpu().numpy() for k, v in t.items()}, 0, device=dev)
        cams = ring_cameras(15, 800, 800, 4, radius=3.0, device=dev)
        bg = torch.zeros(3, device=dev)
        tgt = copy.copy(obj)
        tgt._xyz, tgt._rotation, tgt._features_rest = obj._xyz.detach().clone(), obj._rotation.detach().clone(), obj._features_rest.detach().clone()
        gaussians_rotate(tgt, torch.tensor(rots[planted], dtype=torch.float32, device=dev), fix_center=True)
        targets = render_views(tgt, cams, bg)
        D_sweep = float(np.mean(R_.LAST_NUM_PAIRS_LOG[-len(cams):]))   # (tile, splat) pairs per render of the object (exact-mode renders)
        plan = SweepPlan(obj, cams, targets, bg)                    # eager sizing pass + graph capture (untimed)
        return obj, cams, bg, targets, D_sweep, plan

    def run(ctx):
        obj, cams, bg, targets, D_sweep, plan = ctx
        if world > 1:
            dist.barrier()
        dt, (ids, fit, best) = _event_region(lambda: rotation_sweep(obj, rots[:min(32 * world, len(rots))], cams, targets, bg, plan=plan),
                                             lambda: rotation_sweep(obj, rots, cams, targets, bg, plan=plan))
        dt = _max_over_ranks(dt, cdev, world)
        PairPolicy.reset()
        b = n_obj * 56 + 800 * 800 * 20 + 24 * D_sweep       # B_fwd per render
        rps = len(rots) * len(cams) / dt
        return {"metric": "pose hypotheses/s, 128-rotation sweep (S4: 100k SH0 object x 15 cameras 800x800, config #3)",
                "value": round(len(rots) / dt, 2), "unit": "hypotheses/s", "renders_per_s": round(rps, 1), "seconds_per_sweep": round(dt, 4),
                "n_gpus": world, "scaling": "strong", "stacked_views": plan.stacked is not None, "best_id": best, "planted_id": planted,
                "roofline": {"bound": "hbm", "limited_by": "kernel time of the binning + scoring blend at 5.9 M pairs per hypothesis", "achieved": round(rps * b / 1e9 / world, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": round(rps * b / 1e9 / world / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_render": int(b),
                             "pairs_per_render_D": round(D_sweep)}}
    return guarded_record(prepare, run, cdev)


def secondary_post_refine_objects(dev, cdev, rank, world, iters=24, warm=4):
    """Secondary record, every N: BASELINE config #4, "post_refine_gs.py, 4 objects in parallel on 4 GPUs" - four 100 k
    SH-0 objects at 1600x1200, colours only, masked L1 + SSIM, FusedAdam; object j -> rank j mod N (train.
    post_refine_objects), one all-gather of the refined colours at the end (inside the timed region).  A timed slice
    of `iters` iterations per object stands for the reference's 800 (README.md:153); STRONG scaling: 4 objects
    whatever N.  Equal to the reference's joint refinement only where the objects' screen footprints are disjoint
    (DESIGN.md, section 6); the joint model on one GPU is scripts/measure_configs.py's config #4 line."""
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd.renderer import render as render3d
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    from scorp_amd.train import post_refine_objects
    n_obj, n_pts = 4, 100_000

    def prepare():
        raws = [make_gaussians(n_pts, 0, 50 + k, extent=0.5, log_scale_mean=math.log(0.01)) for k in range(n_obj)]
        for k, r in enumerate(raws):
            r["xyz"] += np.array([(k % 2) * 1.2 - 0.6, (k // 2) * 1.2 - 0.6, 0], np.float32)
        cams = ring_cameras(8, 1600, 1200, 9, device=dev)
        bg, pipe = torch.zeros(3, device=dev), Pipe()
        objs = [GaussianModel.from_raw(r, 0, device=dev) for r in raws]
        mine = list(range(rank, n_obj, world))
        masks = {}
        with torch.no_grad():
            merged = GaussianModel.from_raw({kk: np.concatenate([r[kk] for r in raws]) for kk in raws[0]}, 0, device=dev)
            gts = [render3d(c, merged, pipe, bg)["render"].clamp(0, 1) for c in cams]
            del merged
            for j in mine:
                masks[j] = [(render3d(c, objs[j], pipe, bg)["render_alpha"] > 0.5).float() for c in cams]
            g = torch.Generator(device=dev).manual_seed(7)
            for o in objs:      # the student: perturbed colours (every rank draws the same perturbation)
                o._features_dc.data.add_(0.3 * torch.randn(o._features_dc.shape, device=dev, generator=g))
        return objs, cams, gts, [masks.get(j) for j in range(n_obj)], OptimizationParams()

    def run(ctx):
        objs, cams, gts, alphas, opt = ctx
        if world > 1:
            dist.barrier()
        dt, losses = _event_region(lambda: post_refine_objects(objs, cams, gts, alphas, opt, iterations=warm),   # (allocations, reservation contexts)
                                   lambda: post_refine_objects(objs, cams, gts, alphas, opt, iterations=iters))
        dt = _max_over_ranks(dt, cdev, world)
        PairPolicy.reset()
        ok = all(math.isfinite(v) for ls in losses.values() for v in ls)
        return {"metric": "post-refinement object-iterations/s (config #4: 4 x 100k SH0 objects, 1600x1200, colours only, masked L1+SSIM, FusedAdam)",
                "value": round(n_obj * iters / dt, 1), "unit": "object-iterations/s", "n_gpus": world, "scaling": "strong",
                "iterations_timed_per_object": iters, "seconds_for_800_iterations_of_all_objects": round(800 * dt / iters, 2), "losses_finite": ok}
    return guarded_record(prepare, run, cdev)


def secondary_dp_train(dev, cdev, rank, world, backend):
    """N > 1 only: data-parallel training of ONE scene - the code path with a real exchange step
    (bucketed all-reduce / visibility-sparse reduce-scatter + all-gather of the gradient rows over RCCL)."""
    def prepare():
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import dp_train_rehearsal as dpr
        return dpr, argparse.Namespace(n=200_000, width=1600, height=1200, iters=16)

    def run(ctx):
        dpr, ns = ctx
        return {"metric": "data-parallel training iterations/s, one scene, one view per rank per iteration", "ranks": world,
                "backend": backend, **dpr.run(ns, dev, cdev, rank, world)}
    return guarded_record(prepare, run, cdev)


def launch_ranks(n):
    """`python bench.py --gpus N` from a bare shell (no WORLD_SIZE): start the N ranks as CHILD processes of
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` with the same arguments, relay their output (rank 0
    prints the JSON line) and return the launcher's exit code.  Runs before this process has made any GPU call (no
    torch.cuda.*, no _C.lib()); nothing replaces a running process with another."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)      # >= 200 views after 20 warm-up views
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scene", default="S3")
    ap.add_argument("--cams", type=int, default=0,
                    help="distinct cameras (with resident GT images) cycled per rank; 0 = this rank's share of the whole ring "
                         "(SURVEY 8d: 280 cameras; 23 MB of ground truth each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for rehearsals)")
    ap.add_argument("--single-device", action="store_true",
                    help="rehearsal on a 1-GPU box: every rank uses cuda:0 and collectives go through CPU copies (use with --backend gloo)")
    ap.add_argument("--streams", type=int, default=1,
                    help="views in flight per GPU, each on its own HIP stream (default 1 = one view at a time, as the reference trains)")
    ap.add_argument("--autograd", action="store_true",
                    help="step = render() + fused_l1_ssim_loss() + loss.backward() through torch autograd (the reference's call "
                         "pattern) instead of the single-call scorp_gs3d_train_view; same kernels, more host work per view")
    ap.add_argument("--unfused", action="store_true", help="reference call-site convention: torch activations + cat per view")
    ap.add_argument("--spatial-sort", action="store_true", help="GaussianModel.sort_spatially() first: Gaussians stored along a Z-order curve (not the default)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary record (the 2DGS workload S6) of the default N=1 run")
    ap.add_argument("--lead-in", type=int, default=40, help="untimed views enqueued in front of the warm-up of every timed run (see timed_run)")
    ap.add_argument("--no-twins", action="store_true", help="skip the two extra timed runs (all-fp32 backward, deterministic backward): profiling passes")
    ap.add_argument("--exact-backward", action="store_true", help="(kept for old command lines: the all-fp32 backward IS the headline now) also skips the secondary records")
    ap.add_argument("--split-backward", action="store_true",
                    help="headline = the library's default backward (two-term fp16 split of the pixel->splat reduction, 22 mantissa bits) "
                         "instead of the all-fp32 one")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))      # bare `python bench.py --gpus N`: this process only starts the N ranks (no GPU call made here)
    # ONE line on stdout: whatever else writes to file descriptor 1 from here on - gloo's connection banner, a library's
    # printf, a stray print() - goes to stderr; the JSON line leaves through a private duplicate of the original descriptor.
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    dev_index = 0 if args.single_device else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import datetime
        # (a bounded collective timeout: a rank stuck in a secondary record's exchange errors out instead of hanging the job)
        tmo = datetime.timedelta(minutes=5)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")     # where collective payloads live

    from scorp_amd import _C
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd.renderer import render as render3d
    from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras
    surfels = args.scene == "S6"                      # BASELINE config #5: the 2DGS surfel path
    if surfels:
        from scorp_amd.renderer2d import GaussianModel2D as GaussianModel, render as render2d, fused_surfel_regularizers as surfel_regularizers
        render = render2d
    else:
        render = render3d
    _C.lib()  # fail loudly if the HIP extension is missing

    N, W, H, deg, seed, ncam_total = SCENES[args.scene]
    K = (deg + 1) ** 2
    # rank 0 draws the scene; the others receive it over RCCL (the one collective of this workload, outside the timed region)
    raw = make_gaussians(N, deg, seed, scale_dims=2 if surfels else 3) if rank == 0 else None
    if world > 1:
        from scorp_amd.parallel import broadcast_tensors
        shapes = dict(xyz=(N, 3), scaling=(N, 2 if surfels else 3), rotation=(N, 4), opacity=(N, 1), features_dc=(N, 1, 3), features_rest=(N, K - 1, 3))
        recv = {k: (torch.tensor(raw[k], device=cdev) if rank == 0 else torch.empty(shp, dtype=torch.float32, device=cdev))
                for k, shp in shapes.items()}
        broadcast_tensors(recv, src=0)     # ONE flat 236 MB collective (direct 1 -> N-1 copies over xGMI)
        recv = {k: t.to(dev) for k, t in recv.items()}
        model = GaussianModel(deg, device=dev)
        P = lambda t: torch.nn.Parameter(t.contiguous().requires_grad_(True))
        model._xyz, model._features_dc, model._features_rest = P(recv["xyz"]), P(recv["features_dc"]), P(recv["features_rest"])
        model._scaling, model._rotation, model._opacity = P(recv["scaling"]), P(recv["rotation"]), P(recv["opacity"])
    else:
        model = GaussianModel.from_raw(raw, deg, device=dev)
    model.active_sh_degree = deg
    if args.spatial_sort:
        model.sort_spatially()
    params = [model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity]

    all_cams = ring_cameras(ncam_total, W, H, seed, device=dev)
    n_cams = args.cams if args.cams > 0 else max(ncam_total // world, min(8, ncam_total))
    my_cams = [all_cams[(rank + world * i) % ncam_total] for i in range(n_cams)]   # view i -> rank i mod world
    bg = torch.zeros(3, device=dev)
    pipe = Pipe()
    pipe.fused_activations = not args.unfused

    # resident ground truth + per-camera pair / visibility counts (exact mode, outside the timed region)
    gts, Ds, Nvis = [], [], []
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    with torch.no_grad():
        for cam in my_cams:
            out = render(cam, model, pipe, bg)
            gts.append((out["render"] + 0.05 * torch.randn(out["render"].shape, device=dev, generator=g)).clamp(0, 1))
            Nvis.append(int((out["radii"] > 0).sum()))
    from scorp_amd import rasterizer3d as R
    Ds = list(R.LAST_NUM_PAIRS_LOG[-len(my_cams):])
    D_mean, Nvis_mean = float(np.mean(Ds)), float(np.mean(Nvis))

    fused_view = not surfels and not args.autograd and not args.unfused
    if fused_view:
        from scorp_amd.train_view import train_view
    if args.streams > 1 and not fused_view:
        raise SystemExit("--streams > 1 runs the one-call 3DGS view (scorp_gs3d_train_view): not with --scene S6, --autograd or --unfused")

    # --streams S > 1: S views in flight on S HIP streams (views are independent given the parameters: the batch-of-views
    # form of training, or multi-view evaluation); the default, 1, is the reference's one-view-at-a-time loop
    side_streams = [torch.cuda.Stream() for _ in range(args.streams)] if args.streams > 1 else None

    # `value` is measured with the blend backward's pixel->splat reduction on fp32 MFMAs throughout (SCORP_BACKWARD_EXACT_FP32):
    # every operand at fp32, as in the reference's arithmetic.  The library's DEFAULT form carries the two operands of that
    # reduction as two fp16 terms each (22 mantissa bits, exact products, fp32 accumulation; parity-tested against the exact
    # form and the oracle): faster, reported as `value_split22`, and the headline only with --split-backward.
    headline_exact = fused_view and not args.split_backward
    if headline_exact or args.exact_backward:
        R._tls.backward_flags = _C.BACKWARD_EXACT_FP32     # every forward (and one-call view) from here on asks for the all-fp32 backward

    def step(i):
        cam, gt = my_cams[i % len(my_cams)], gts[i % len(my_cams)]
        if side_streams is not None:
            with torch.cuda.stream(side_streams[i % args.streams]):
                loss = train_view(cam, model, pipe, bg, gt, 0.2)["loss"]
                for p in params:
                    p.grad = None
            return loss
        if fused_view:      # render + loss + backward enqueued by one library call (scorp_gs3d_train_view): same kernels
            loss = train_view(cam, model, pipe, bg, gt, 0.2)["loss"]
            for p in params:
                p.grad = None
            return loss
        out = render(cam, model, pipe, bg)
        loss = fused_l1_ssim_loss(out["render"], gt, 0.2)
        if surfels:                                   # train_2dgs.py:142-150: normal consistency + depth distortion
            nl, dl = surfel_regularizers(out, 0.05, 100.0)
            loss = loss + nl + dl
        loss.backward()
        for p in params:
            p.grad = None
        return loss

    PairPolicy.mode, PairPolicy.reserve = "reserve", int(max(Ds) * 1.25) + 1024
    # probe (setup, untimed, before the warm-up): a few views with EVERY kernel bracketed by hipEvents give the per-kernel
    # table and tell which kernel dominates
    kern_all, dominant, reconcile = {}, None, None
    if not args.no_kernel_events:
        for i in range(3):       # first launches of the step's kernels (code objects, scratch): not in the per-kernel table
            step(i)
        PairPolicy.drain()
        _C.prof_enable(True)
        for i in range(max(3, min(args.warmup, 8))):
            step(i)
        PairPolicy.drain()
        torch.cuda.synchronize()
        kern_all = _C.prof_collect()
        dominant = max(kern_all, key=lambda k: kern_all[k][0]) if any(c for _, c in kern_all.values()) else None
        # Reconciliation of the per-kernel table with the step: the same few views once more with every
        # kernel bracketed and ONE event pair around them, then with nothing bracketed.  kernel sum + gaps = bracketed view;
        # bracketed view - plain view = what the brackets themselves cost (an event pair is a few microseconds of stream time,
        # part of it inside the pair); the plain view is the timed region's ms_per_step up to the run's noise.
        n_rec = 8
        _C.prof_enable(True)
        t_br, _ = _event_region(lambda: [step(i) for i in range(4)], lambda: [step(4 + i) for i in range(n_rec)])
        PairPolicy.drain()
        kern_rec = _C.prof_collect()
        _C.prof_enable(False)
        t_pl, _ = _event_region(lambda: [step(i) for i in range(4)], lambda: [step(4 + i) for i in range(n_rec)])
        PairPolicy.drain()
        n_views_rec = n_rec + 4     # (the lead-in views were bracketed too)
        reconcile = {"kernel_sum_us": round(sum(ms for ms, c in kern_rec.values() if c) / n_views_rec * 1e3, 1),
                     "brackets_per_view": round(sum(c for _, c in kern_rec.values()) / n_views_rec, 1),
                     "bracketed_view_us": round(t_br / n_rec * 1e6, 1), "plain_view_us": round(t_pl / n_rec * 1e6, 1)}
        reconcile["gaps_between_brackets_us"] = round(reconcile["bracketed_view_us"] - reconcile["kernel_sum_us"], 1)
        reconcile["bracketing_cost_us_per_view"] = round(reconcile["bracketed_view_us"] - reconcile["plain_view_us"], 1)
        reconcile["bracketing_cost_us_per_bracket"] = round(reconcile["bracketing_cost_us_per_view"] / max(reconcile["brackets_per_view"], 1), 2)

    def timed_run(n_warm, n_steps, bracket=None, lead_in=args.lead_in):
        """n_warm untimed views, then EXACTLY n_steps views between two hipEvents recorded on the launch stream directly
        behind the warm-up: no host synchronisation opens the timed region (a synchronise leaves the chip idle for a
        moment and the first ~20 views after it run 5-20 % slow, scripts/dev/ramp.py - at the driver's --steps 20 that
        was the whole region).  For the same reason `lead_in` more views are enqueued in front of the warm-up, behind the
        synchronise that separates this run from the probe / the previous run: with a short --warmup the warm-up itself
        would otherwise sit on the ramp (measured at --steps 20 --warmup 5 on one box: 1 315-1 320 views/s with 12 lead-in
        views, 1 352-1 355 with 40, 1 355-1 356 with 100; the default 200-step run gives the same with any of them: 40 it is,
        30 ms per timed run).  barrier + synchronize bracket the whole; the host clock over the region (enqueue
        start -> synchronize) is kept as a cross-check.  Returns (event s, host s, host enqueue s, last loss)."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        if not args.no_kernel_events:
            _C.prof_enable(True, only=[])                      # nothing bracketed during the lead-in and the warm-up
        for i in range(lead_in + n_warm):
            step(i)
        if not args.no_kernel_events and bracket:
            _C.check(_C.lib().scorp_prof_select(bracket), "scorp_prof_select")   # only the dominant kernel, from here on
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        h0 = time.perf_counter()
        loss_ = None
        for i in range(n_steps):
            loss_ = step(n_warm + i)
        h_enq = time.perf_counter() - h0
        ev1.record()
        PairPolicy.drain()
        torch.cuda.synchronize()
        h1 = time.perf_counter() - h0
        if world > 1:
            dist.barrier()
        return ev0.elapsed_time(ev1) * 1e-3, h1, h_enq, loss_

    dom_mask = 0
    if dominant:
        names_ = [_C.lib().scorp_prof_kernel_name(k).decode() for k in range(_C.lib().scorp_prof_num_kernels())]
        dom_mask = 1 << names_.index(dominant)
    dt, dt_host, t_host, loss = timed_run(args.warmup, args.steps, dom_mask)
    kern = {} if args.no_kernel_events else _C.prof_collect()
    _C.prof_enable(False)
    # the same step with the OTHER form of the blend backward's pixel->splat reduction (the two-term fp16 split if the headline
    # is all-fp32, and the other way round), same protocol: the headline next to its twin
    dt_twin = dt_det = None
    if fused_view and side_streams is None and not args.no_twins:
        prev_flags = getattr(R._tls, "backward_flags", 0)
        R._tls.backward_flags = 0 if (prev_flags & _C.BACKWARD_EXACT_FP32) else _C.BACKWARD_EXACT_FP32
        try:
            dt_twin = timed_run(min(args.warmup, 10), args.steps)[0]
            # ... and with SCORP_BACKWARD_DETERMINISTIC: no float atomics, plain partial rows + an ordered per-Gaussian sum
            R._tls.backward_flags = _C.BACKWARD_DETERMINISTIC
            dt_det = timed_run(min(args.warmup, 5), args.steps)[0]
        finally:
            R._tls.backward_flags = prev_flags
    exact_is_headline = bool(getattr(R._tls, "backward_flags", 0) & _C.BACKWARD_EXACT_FP32)
    dt_exact, dt_split = (None, dt_twin) if exact_is_headline else (dt_twin, None)
    R._tls.backward_flags = 0     # the extras and the secondary records below run the library's defaults
    # extra (not part of `value`): what an UNMODIFIED SCORP script gets (north_star: "train_3dgs.py ... run unchanged") - the
    # reference's own call pattern per view (train_3dgs.py:94-152): render(cam, gaussians, pipe, bg) with the reference's
    # PipelineParams (arguments/__init__.py: convert_SHs_python, compute_cov3D_python, debug - nothing of this package's), the
    # exact pair count (one 8-byte device-to-host read per view), l1_loss / ssim by the reference's names
    # (scorp_amd/loss.py restates gs3dgs/utils/loss_utils.py:17-73), loss.backward() through torch autograd, the library's
    # default rasterizer backward.  What the package does by itself for such a caller: render() hands the stock
    # GaussianModel's raw leaves to the kernels (no torch activations / cat), `ssim` answers from the HIP loss kernels.
    # `torch_activations_and_ssim`: the same step with both switched off (pipe.fused_activations = False, ssim_torch) -
    # torch sigmoid / exp / normalize / cat and five depthwise 11x11 conv2d per view: round 6's first definition of the figure.
    dropin = None
    if world == 1 and not surfels and side_streams is None and not args.no_secondary:
        try:
            from scorp_amd.loss import l1_loss as ref_l1, ssim as ref_ssim, ssim_torch

            class RefPipe:          # the reference's PipelineParams
                convert_SHs_python = False
                compute_cov3D_python = False
                debug = False

            class TorchPipe(RefPipe):
                fused_activations = False
            PairPolicy.mode = "exact"

            def step_dropin(i, pipe_=RefPipe, ssim_fn=ref_ssim):
                cam, gt = my_cams[i % len(my_cams)], gts[i % len(my_cams)]
                out = render3d(cam, model, pipe_, bg)
                img = out["render"]
                loss_ = (1.0 - 0.2) * ref_l1(img, gt) + 0.2 * (1.0 - ssim_fn(img, gt))
                loss_.backward()
                for p in params:
                    p.grad = None
            n_d = max(8, min(args.steps, 24))
            dt_d, _ = _event_region(lambda: [step_dropin(i) for i in range(4)], lambda: [step_dropin(4 + i) for i in range(n_d)])
            dt_t, _ = _event_region(lambda: [step_dropin(i, TorchPipe, ssim_torch) for i in range(2)],
                                    lambda: [step_dropin(2 + i, TorchPipe, ssim_torch) for i in range(8)])
            dropin = {"value": round(n_d / dt_d, 1), "ms_per_step": round(dt_d / n_d * 1e3, 3), "steps": n_d,
                      "step": "reference call pattern: render(cam, gaussians, pipe, bg) with the reference's PipelineParams and exact pair "
                              "count (8-byte D2H per view), l1_loss + ssim by the reference's names, autograd backward; library-default "
                              "rasterizer backward (render() takes the stock model's raw leaves, ssim runs on the HIP loss kernels)",
                      "torch_activations_and_ssim": {"value": round(8 / dt_t, 1), "ms_per_step": round(dt_t / 8 * 1e3, 3),
                                                     "step": "the same with torch sigmoid / exp / normalize / cat and ssim as five "
                                                             "depthwise conv2d (MIOpen)"}}
        except Exception as e:   # noqa: BLE001   (an extra: never costs the headline)
            dropin = {"error": f"{type(e).__name__}: {e}"[:200]}
        finally:
            PairPolicy.mode = "reserve"
            for p in params:
                p.grad = None
    # extra (not part of `value`): forward-only render rate, the unit of the alignment sweep / test-view rendering
    nf = max(args.steps // 2, 1)
    torch.cuda.synchronize()
    tf0 = time.perf_counter()
    with torch.no_grad():
        for i in range(nf):
            render(my_cams[i % len(my_cams)], model, pipe, bg)
    PairPolicy.drain()
    torch.cuda.synchronize()
    fwd_only = nf / (time.perf_counter() - tf0)
    # extra (not part of `value`): the same step with TWO views in flight on two HIP streams - the batch-of-views form of
    # training (data-parallel ranks sharing a GPU, multi-view evaluation): the second view's kernels fill the SIMDs that
    # the first one's latency-bound kernels (binning, loss) and kernel tails leave idle
    in_flight2 = None
    if fused_view and side_streams is None and not args.no_secondary:
        pair = [torch.cuda.Stream(), torch.cuda.Stream()]
        for st_ in pair:
            st_.wait_stream(torch.cuda.current_stream())

        def step2(i):
            with torch.cuda.stream(pair[i & 1]):
                train_view(my_cams[i % len(my_cams)], model, pipe, bg, gts[i % len(my_cams)], 0.2)
                for p in params:
                    p.grad = None
        for i in range(8):
            step2(i)
        PairPolicy.drain()
        torch.cuda.synchronize()
        t20 = time.perf_counter()
        for i in range(args.steps):
            step2(i)
        PairPolicy.drain()
        torch.cuda.synchronize()
        in_flight2 = args.steps / (time.perf_counter() - t20)
        if world > 1:
            t2 = torch.tensor([in_flight2], device=cdev, dtype=torch.float64)
            dist.all_reduce(t2, op=dist.ReduceOp.SUM)
            in_flight2 = float(t2.item())
    PairPolicy.mode = "exact"
    # work statistics of view 0 (SURVEY §8d asks for an honest pixel-splat figure next to HBM): (8x8 block, splat)
    # iterations of the two blend kernels, each of which evaluates 64 pixel-splat pairs
    work = None
    if rank == 0 and not surfels:
        import ctypes
        R.KEEP_LAST_FORWARD = True
        render(my_cams[0], model, pipe, bg)      # grad-enabled: the forward leaves its per-block statistics
        R.KEEP_LAST_FORWARD = False
        st, n_, w_, h_ = R.LAST_FORWARD
        o3 = (ctypes.c_uint64 * 3)()
        _C.check(_C.lib().scorp_gs3d_debug_work(st.data_ptr(), n_, w_, h_, ctypes.byref(o3), R._stream()), "scorp_gs3d_debug_work")
        R.LAST_FORWARD = None
        work = {"forward_block_splat_iterations": int(o3[0]), "backward_block_splat_iterations": int(o3[1]), "blocks_8x8": int(o3[2])}
    if world > 1:
        tt = torch.tensor([dt, dt_host, dt_exact or 0.0, dt_det or 0.0, dt_split or 0.0], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, dt_host = float(tt[0]), float(tt[1])
        dt_exact = float(tt[2]) if dt_exact is not None else None
        dt_det = float(tt[3]) if dt_det is not None else None
        dt_split = float(tt[4]) if dt_split is not None else None
        ll = torch.tensor([float(loss.detach())], device=cdev)
        gathered = [torch.zeros_like(ll) for _ in range(world)]
        dist.all_gather(gathered, ll)     # gather of per-rank results (scalars)

    # view 0 of the UNTRAINED parameters for the full-size parity figures (the training extra below moves the model)
    hip_view0 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not surfels:
        with torch.no_grad():
            hip_view0 = render(my_cams[0], model, pipe, bg)["render"].cpu().numpy()
    # extra (not part of `value`), N = 1: FULL training iterations on the same model and cameras - the one-call view, the
    # per-view densification statistics (scorp_densification_stats) and the guarded FusedAdam step over all 59 parameters
    # per Gaussian (train_3dgs.py:74-193 / train_2dgs.py without the densification itself, which runs every 100th iteration)
    train_its = None
    if world == 1 and fused_view and side_streams is None and not args.no_secondary and not args.exact_backward:
        try:
            from scorp_amd.gaussian_model import OptimizationParams, OptimizationParams2D
            from scorp_amd.train import training_iteration
            PairPolicy.mode = "reserve"
            opt_ = OptimizationParams2D() if surfels else OptimizationParams()
            opt_.densify_from_iter, opt_.opacity_reset_interval, opt_.random_background = 1 << 30, 1 << 30, False
            model.optimizer = None
            model.training_setup(opt_)
            kw_ = dict(surfels=True) if surfels else {}
            n_it = 60   # (a fixed count, 65 ms: the driver's --steps 20 form would time 20 ms of it)

            def it_(i):
                k_ = i % len(my_cams)
                training_iteration(model, my_cams[k_], gts[k_], opt_, pipe, bg, i + 1, scene_extent=3.0, fused_view=True, **kw_)
            for i in range(12):
                it_(i)
            PairPolicy.drain()
            e0_, e1_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0_.record()
            for i in range(n_it):
                it_(12 + i)
            e1_.record()
            PairPolicy.drain()
            torch.cuda.synchronize()
            train_its = round(n_it / (e0_.elapsed_time(e1_) * 1e-3), 1)
        except Exception as e:   # (an extra: never costs the headline)
            train_its = f"{type(e).__name__}: {e}"[:200]
        finally:
            PairPolicy.mode = "exact"
            model.optimizer = None
            for p in params:
                p.grad = None
    # secondary records (guarded_record: a failure in one of them is reported inside the record, on every rank, and never
    # costs the headline): every N the 128-rotation sweep (the workload north_star's 8-GPU scaling target is set on) and
    # the object-sharded post-refinement; N > 1 the data-parallel training of one scene; N = 1 the 2DGS workload S6
    secondary = {}
    if args.scene == "S3" and not args.no_secondary and not args.exact_backward and args.streams == 1:
        secondary["sweep_128"] = secondary_sweep(dev, cdev, rank, world)
        secondary["post_refine_4obj"] = secondary_post_refine_objects(dev, cdev, rank, world)
        if world > 1:
            secondary["dp_train"] = secondary_dp_train(dev, cdev, rank, world, args.backend)
    parity_failed = False
    if rank == 0:
        views = args.steps * world
        value = views / dt
        HW = W * H
        roof, kernels = None, {}
        if kern:
            merged = dict(kern_all)
            merged.update({k: v for k, v in kern.items() if v[1]})   # dominant kernel: live numbers of the timed region
            for name, (ms, cnt) in merged.items():
                if cnt:
                    b_ = kernel_algorithmic_bytes(name, N, Nvis_mean, K, HW, D_mean)
                    kernels[name] = [round(ms / cnt * 1e3, 1), round(b_ / (ms / cnt * 1e-3) / 1e9)]     # [avg us, algorithmic GB/s]
            dom = dominant if dominant in kernels else max(kernels, key=lambda k: kernels[k][0])
            lib_sha = _C.lib().scorp_source_sha().decode()
            traffic, stale = None, []
            tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-measured HBM bytes per launch (profiles/README.md)
            if os.path.exists(tpath):
                tj = json.load(open(tpath))
                if tj.get("source_sha") == lib_sha:
                    traffic = tj.get(args.scene, {}).get(dom)
                else:
                    stale.append("traffic.json")
            # The limiter the HBM figure cannot show: the blend kernels are bound by VALU issue.  Per (8x8 block,
            # splat) iteration the hot block's instruction mix (profiles/valu_mix.json, scripts/isa_mix.py) priced with the
            # issue costs MEASURED by scripts/mb_valu_peak.hip gives the launch time with the VALU issuing back to back.
            valu = None
            mpath = os.path.join(ROOT, "profiles", "valu_mix.json")
            if work and os.path.exists(mpath):
                mj = json.load(open(mpath))
                if mj.get("source_sha") == lib_sha:
                    valu = {}
                    for kn, its in (("blend_forward", work["forward_block_splat_iterations"]), ("blend_backward", work["backward_block_splat_iterations"])):
                        if kn in kernels and kn in mj:
                            cyc = mj[kn]["valu_cycles_per_hit"] + mj[kn]["mfma_cycles_per_hit"]
                            bound_us = its * cyc / 1024 / 2.4e9 * 1e6
                            valu[kn] = {"insts_per_iteration": mj[kn]["valu_insts_per_hit"], "cycles_per_iteration": round(cyc, 1),
                                        "bound_us": round(bound_us, 1), "issue_slot_frac": round(bound_us / kernels[kn][0], 3)}
                else:
                    stale.append("valu_mix.json")
            alg_dom = kernel_algorithmic_bytes(dom, N, Nvis_mean, K, HW, D_mean)
            roof = dict(bound="hbm", limited_by="vector instruction issue (valu)", kernel=dom, achieved=kernels[dom][1], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(alg_dom / (kernels[dom][0] * 1e-6) / 1e9 / HBM_PEAK_GBS, 5), traffic=traffic,
                        avg_launch_us=kernels[dom][0], algorithmic_bytes=int(alg_dom), valu=valu)
            if stale:
                roof["stale_profiles"] = stale      # collected on other kernel sources than the loaded library's (scorp_source_sha)
        B_view = N * 720 + HW * 40 + 28 * D_mean
        P_f = 64 * work["forward_block_splat_iterations"] if work else None
        P_b = 64 * work["backward_block_splat_iterations"] if work else None
        line = {
            "metric": "fwd+bwd views/sec @1M Gaussians 1600x1200 SH3" if args.scene == "S3" else f"fwd+bwd views/sec ({args.scene}{', 2DGS surfels' if surfels else ''})",
            "unit": "views/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.scene}: {N} Gaussians, {W}x{H}, SH degree {deg}, ring cameras (SURVEY 8d)",
                       "step": "steady-state: render fwd + 0.8*L1+0.2*(1-SSIM) + backward to 59 params/Gaussian, no optimizer step; "
                               + ("one scorp_gs3d_train_view call per view" if fused_view else "render + fused loss + autograd backward"),
                       "views_per_rank": args.steps, "distinct_cameras_per_rank": len(my_cams), "views_in_flight": args.streams,
                       "pairs_per_view_D": round(D_mean), "D_over_N": round(D_mean / N, 3), "visible": round(Nvis_mean),
                       "pixel_splat_pairs_P": P_f, "pixel_splat_pairs_P_backward": P_b, "parallelism": f"view-sharded replicas x{world}"},
            "timing": {"clock": "hipEvents", "probe_views": max(3, min(args.warmup, 8)), "lead_in_views": args.lead_in,
                       "host_clock_ms_per_step": round(dt_host / args.steps * 1e3, 4), "host_enqueue_ms_per_step": round(1e3 * t_host / args.steps, 4)},
            "precision": "f32 throughout (blend backward's pixel->splat reduction on fp32 MFMAs)" if exact_is_headline
                         else "f32; backward pixel->splat reduction: 2-term fp16 split MFMA (22 bits), fp32 accumulate",
            "library_default_backward": "2-term fp16 split (value_split22); the headline asks for SCORP_BACKWARD_EXACT_FP32 per view",

            "kernels_us_GBs": kernels,
            "kernel_table_reconciliation": reconcile,
            "pairs_per_s": (_sig(value * (P_f + P_b), 4) if work else None),
            "forward_only_views_per_s_per_gpu": round(fwd_only, 1),
            "views_per_s_two_in_flight": None if in_flight2 is None else round(in_flight2, 1),
            "train_iterations_per_s": train_its,
            "view_hbm": {"algorithmic_bytes_per_view": int(B_view), "achieved_GBs": round(value / world * B_view / 1e9, 1),
                         "frac_of_8TBs": round(value / world * B_view / 8e12, 5)},
        }
        cpu_rec = None
        if not args.no_cpu_baseline and world == 1:
            cpu_rec, orc = cpu_baseline(raw, my_cams[0].to("cpu"), deg, W, H)
            line["parity"] = small_parity(dev)
            my_cams[0].to(dev)     # (.to() moves in place)
            if not surfels:   # the same full-size view, HIP against the CPU oracle the baseline just rendered
                hip = hip_view0
                mse = float(((hip - orc.color) ** 2).mean())
                # ... and the gradients of the oracle's backward pass (same inputs, same upstream gradient), per tensor
                # relative L1 (north_star's norm; asserted < 1e-4 in tests/test_fullsize_gpu.py) and max-norm, through the
                # reference call convention (activated inputs)
                from scorp_amd.refcall import render3d_reference_call as hip_render
                kw_o, w_o, g_o = orc.full_size_case
                out_h, t_h = hip_render(kw_o, dev)
                (out_h[0] * torch.tensor(w_o, device=dev)).sum().backward()
                errs = {nm: _grad_errors(t_h[nm].grad.detach().cpu().numpy(), g_o[nm]) for nm in GRAD_NAMES}
                line["parity"]["full_size"] = dict(l1=_sig(np.abs(hip - orc.color).mean()), max_abs=_sig(np.abs(hip - orc.color).max()),
                                                   psnr_db=(99.0 if mse == 0 else round(float(10 * math.log10(1.0 / mse)), 1)),
                                                   grad_rel_l1={k: _sig(v[1]) for k, v in errs.items()},
                                                   grad_max_rel_err={k: _sig(v[0]) for k, v in errs.items()})
                # north_star's tolerance, checked here as well as in tests/test_fullsize_gpu.py: the line says so itself
                line["parity_ok"] = bool(line["parity"]["full_size"]["l1"] < 1e-4 and all(v[1] < 1e-4 for v in errs.values()))
                if not line["parity_ok"]:
                    print("[bench] full-size parity outside 1e-4: " + json.dumps(line["parity"]["full_size"]), file=sys.stderr)
        if world == 1 and args.scene == "S3" and not args.no_secondary and not args.exact_backward:
            try:
                secondary["S6"] = secondary_s6(dev, parity=not args.no_cpu_baseline)
            except Exception as e:   # noqa: BLE001
                secondary["S6"] = {"error": f"{type(e).__name__}: {e}"}
        if secondary:
            line["secondary"] = secondary
        # the figures the record is read for come LAST (a log tail keeps the end of the line)
        line["ms_per_step"] = round(dt / args.steps * 1e3, 4)
        line["value"] = round(value, 3)
        line["value_exact_fp32"] = round(value, 3) if exact_is_headline else (None if dt_exact is None else round(views / dt_exact, 3))
        line["ms_per_step_exact_fp32"] = line["ms_per_step"] if exact_is_headline else (None if dt_exact is None else round(dt_exact / args.steps * 1e3, 4))
        line["value_split22"] = (None if dt_split is None else round(views / dt_split, 3)) if exact_is_headline else round(value, 3)
        line["ms_per_step_split22"] = (None if dt_split is None else round(dt_split / args.steps * 1e3, 4)) if exact_is_headline else line["ms_per_step"]
        line["value_dropin"] = None if dropin is None else dropin.get("value")
        line["dropin"] = dropin
        line["value_deterministic_backward"] = None if dt_det is None else round(views / dt_det, 3)
        line["ms_per_step_deterministic_backward"] = None if dt_det is None else round(dt_det / args.steps * 1e3, 4)
        line["roofline"] = roof
        if cpu_rec is not None:
            line["cpu_baseline"] = cpu_rec
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(line, separators=(",", ":")) + "\n").encode())
        parity_failed = line.get("parity_ok") is False
    if world > 1:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:   # noqa: BLE001   (the line is out; a broken group must not turn the run into a failure)
            print(f"[bench] process group teardown: {type(e).__name__}: {e}", file=sys.stderr)
    if parity_failed:
        sys.exit(3)     # the line is out (with "parity_ok": false); a run whose renders miss the oracle is not a success


if __name__ == "__main__":
    main()
