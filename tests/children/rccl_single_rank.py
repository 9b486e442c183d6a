"""Child process of tests/test_rccl_gpu.py: a process group of ONE rank on backend "nccl" (= RCCL), with
scorp_amd.parallel.SINGLE_RANK_COLLECTIVES so that every helper issues its collectives on device tensors - trivial
exchanges, the real code path (init_process_group(device_id=...), broadcast, all_gather_into_tensor,
reduce_scatter_tensor, all_reduce) - and one data-parallel training iteration, the sharded sweep and the object-sharded
refinement on top of them.  Prints RCCL_SINGLE_RANK_OK on success."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SCORP_SINGLE_RANK_COLLECTIVES"] = "1"
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import datetime

import numpy as np
import torch
import torch.distributed as dist


def main():
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev, timeout=datetime.timedelta(minutes=2))
    from scorp_amd import parallel as P
    assert P.collective() and P.world() == (0, 1)
    g = torch.Generator(device=dev).manual_seed(3)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)

    # --- the helpers, each through its collective ---
    t = {"xyz": R(1000, 3), "features_rest": R(1000, 15, 3), "opacity": R(1000, 1)}
    before = {k: v.clone() for k, v in t.items()}
    P.broadcast_tensors(t, src=0)
    assert all(torch.equal(t[k], before[k]) for k in t)
    vals = R(5, 2)
    ids, got = P.gather_results([0, 1, 2, 3, 4], vals, n_total=5)                 # ONE all_gather_into_tensor
    assert ids.tolist() == [0, 1, 2, 3, 4] and torch.equal(got, vals)
    ids, got = P.gather_results([3, 1, 2], vals[:3])                              # ragged form: counts first
    assert ids.tolist() == [1, 2, 3] and torch.equal(got, vals[:3][[1, 2, 0]])
    rows = {0: R(7, 3), 1: R(4, 3)}
    out = P.gather_rows(rows, 2, [7, 4])
    assert torch.equal(out[0], rows[0]) and torch.equal(out[1], rows[1])
    assert P.all_ok(True, dev) and not P.all_ok(False, dev)
    N = 501
    shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 3), (N, 4)]
    vis = (torch.arange(N, device=dev) % 3) != 0
    for mode in ("dense", "packed", "auto"):
        params = [torch.nn.Parameter(R(*s)) for s in shapes]
        for p in params:
            gr = R(*p.shape)
            gr.view(N, -1)[~vis] = 0
            p.grad = gr
        want = [p.grad.clone() for p in params]
        if mode == "dense":
            P.average_gradients(params, bucket_bytes=1 << 12)                      # bucketed all_reduce
        else:                                                                      # reduce_scatter_tensor + all_gather_into_tensor
            moved = P.average_gradients_sparse(params, vis, dense_above=1.0 if mode == "packed" else 0.6)
            assert moved == int(vis.sum())
        assert all(torch.equal(p.grad, w) for p, w in zip(params, want)), mode

    # --- the gradient arena: reduce_scatter_tensor + all_gather_into_tensor on the two parts of one flat buffer, in place ---
    params = [torch.nn.Parameter(R(*s_)) for s_ in shapes]
    arena = P.GradArena(params)
    want = []
    for p_, v_ in zip(params, arena.views):
        v_.copy_(R(*p_.shape))
        p_.grad = v_
        want.append(v_.clone())
    arena.attach()
    arena.average()
    assert all(torch.equal(p_.grad, w_) and p_.grad.data_ptr() == v_.data_ptr() for p_, w_, v_ in zip(params, want, arena.views))

    # --- one scene trained data-parallel on the group of one == the same loop without a group's help ---
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.rasterizer3d import PairPolicy, backward_precision
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    from scorp_amd.train import PipelineParams, post_refine_objects, render_views_gt, train
    raw = make_gaussians(20_000, 3, 2, log_scale_mean=math.log(0.02))
    cams = ring_cameras(4, 320, 240, 2, device=dev)
    teacher = GaussianModel.from_raw(raw, 3, device=dev)
    teacher.active_sh_degree = 3
    gts = render_views_gt(teacher, cams)
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")

    def run(dp, sparse, densify):
        m = GaussianModel.from_raw(raw, 3, device=dev)
        m.active_sh_degree = 3
        m._features_dc.data.add_(0.2)
        opt = OptimizationParams()
        opt.random_background = False
        # densify: a densify step inside the six iterations (statistics all-reduced in front of it, positions drawn from
        # the per-iteration seed the data-parallel loop sets)
        opt.densify_from_iter, opt.densification_interval = (2, 3) if densify else (10_000, 10_000)
        PairPolicy.reset()
        P.SINGLE_RANK_COLLECTIVES = dp
        try:
            with backward_precision("deterministic"):     # (no float-atomic noise: the runs can be compared closely)
                losses = train(m, cams, gts, opt, PipelineParams(), iterations=6, data_parallel=dp, fused_view=True,
                               sparse_gradients=sparse, seed=5)
        finally:
            P.SINGLE_RANK_COLLECTIVES = True
        assert all(math.isfinite(v) for v in losses)
        return [getattr(m, n).detach().clone() for n in names]

    def same(x, y):
        for a, b in zip(x, y):
            assert a.shape == b.shape and float((a - b).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1.0)
    same(run(False, False, False), run(True, False, False))     # a group of one averages to itself
    same(run(True, False, True), run(True, True, True))         # dense == visibility-sparse, through a densify step
    PairPolicy.reset()

    # --- the sharded rotation sweep (one fixed-size all-gather) and the object-sharded refinement (gather_rows) ---
    from scorp_amd.align import render_views, rotation_sweep
    rots = np.load(os.path.join(ROOT, "tests", "golden", "rotations_128.npz"))["rotations"][:8]
    obj = GaussianModel.from_raw(make_gaussians(5_000, 0, 4, extent=0.8, log_scale_mean=math.log(0.03)), 0, device=dev)
    cams3 = ring_cameras(3, 128, 128, 4, radius=3.0, device=dev)
    bg = torch.zeros(3, device=dev)
    targets = render_views(obj, cams3, bg)
    ids, fit, best = rotation_sweep(obj, rots, cams3, targets, bg)
    assert ids.tolist() == list(range(8)) and fit.shape == (8, 1) and 0 <= best < 8
    PairPolicy.reset()
    # ... in the form bench.py's 8-GPU record takes: a plan built beforehand (stacked views, RESERVED pair buffers: no host
    # synchronisation per render), the sharded sweep through the group's all-gather, against this rank scoring everything
    # itself without the group's help
    from scorp_amd.align import SweepPlan
    plan = SweepPlan(obj, cams3, targets, bg)
    assert plan.stacked is not None
    ids2, fit2, best2 = rotation_sweep(obj, rots, cams3, targets, bg, plan=plan)
    P.SINGLE_RANK_COLLECTIVES = False
    try:
        ids3, fit3, best3 = rotation_sweep(obj, rots, cams3, targets, bg, plan=plan, shard=False)
    finally:
        P.SINGLE_RANK_COLLECTIVES = True
    assert ids2.tolist() == ids3.tolist() and best2 == best3 == best
    assert float((fit2.to(fit3.device) - fit3).abs().max()) < 1e-6      # (the score is a float-atomic sum: not bit-stable)
    assert float((fit2.to(fit.device) - fit).abs().max()) < 1e-6
    PairPolicy.reset()
    objs = [GaussianModel.from_raw(make_gaussians(3_000, 0, 60 + k, extent=0.5, log_scale_mean=math.log(0.03)), 0, device=dev) for k in range(2)]
    cams4 = ring_cameras(2, 160, 128, 9, device=dev)
    with torch.no_grad():
        from scorp_amd.renderer import render
        gts4 = [render(c, objs[0], PipelineParams(), bg)["render"].clamp(0, 1) for c in cams4]
        alphas = [[(render(c, o, PipelineParams(), bg)["render_alpha"] > 0.5).float() for c in cams4] for o in objs]
    losses = post_refine_objects(objs, cams4, gts4, alphas, OptimizationParams(), iterations=3)
    assert sorted(losses) == [0, 1] and all(math.isfinite(v) for ls in losses.values() for v in ls)
    PairPolicy.reset()
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_SINGLE_RANK_OK", flush=True)


if __name__ == "__main__":
    main()
