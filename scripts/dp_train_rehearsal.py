#!/usr/bin/env python3
"""Data-parallel training of ONE scene on G ranks (train_3dgs.py:56-193 with one view per rank):
iterations/s of scorp_amd.train.train(data_parallel=True, fused_view=True), dense and visibility-sparse gradient average.

  one GPU, two ranks sharing it (rehearsal; collectives through gloo / host copies):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
        scripts/dp_train_rehearsal.py --backend gloo --single-device
  G GPUs (RCCL over xGMI):  ... --nproc-per-node G scripts/dp_train_rehearsal.py
Rank 0 prints one JSON line.  The replicas must stay bit-identical: a parameter checksum is compared across ranks."""
import argparse, json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def run(args, dev, cdev, rank, world):
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.rasterizer3d import PairPolicy
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    from scorp_amd.train import PipelineParams, render_views_gt, train
    raw = make_gaussians(args.n, 3, 2, log_scale_mean=math.log(0.006))
    teacher = GaussianModel.from_raw(raw, 3, device=dev)
    teacher.active_sh_degree = 3
    cams = ring_cameras(16, args.width, args.height, 2, device=dev)
    gts = render_views_gt(teacher, cams)
    out = {}
    trace = []
    if os.environ.get("SCORP_DP_TRACE"):   # per-iteration wall times (synchronised) of rank 0, printed to stderr
        import scorp_amd.train as T
        inner = T.training_iteration

        def timed(*a, **k):
            torch.cuda.synchronize(); t_ = time.perf_counter()
            r = inner(*a, **k)
            torch.cuda.synchronize(); trace.append(round(1e3 * (time.perf_counter() - t_), 1))
            return r
        T.training_iteration = timed
        for name in ("average_gradients", "average_gradients_sparse", "_shared_overflow"):   # the exchange steps on their own
            def wrap(fn, name=name):
                def w(*a, **k):
                    torch.cuda.synchronize(); t_ = time.perf_counter()
                    r = fn(*a, **k)
                    torch.cuda.synchronize(); d_ = 1e3 * (time.perf_counter() - t_)
                    if d_ > 30.0:
                        trace.append(f"{name}:{d_:.0f}")
                    return r
                return w
            setattr(T, name, wrap(getattr(T, name)))
    for sparse in ((True, False) if os.environ.get("SCORP_DP_SPARSE_FIRST") else (False, True)):
        m = GaussianModel.from_raw(raw, 3, device=dev)
        m.active_sh_degree = 3
        m._features_dc.data.add_(0.2)
        opt = OptimizationParams()
        opt.random_background = False
        opt.densify_from_iter, opt.densification_interval = 10_000, 10_000     # steady-state iterations
        PairPolicy.reset()
        train(m, cams, gts, opt, PipelineParams(), iterations=8, data_parallel=True, fused_view=True, sparse_gradients=sparse)
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        losses = train(m, cams, gts, opt, PipelineParams(), iterations=args.iters, data_parallel=True, fused_view=True,
                       sparse_gradients=sparse, seed=1)
        torch.cuda.synchronize(); dist.barrier()
        dt = time.perf_counter() - t0
        chk = torch.stack([p.detach().double().sum() for p in (m._xyz, m._features_dc, m._features_rest, m._opacity, m._scaling, m._rotation)]).to(cdev)
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        out["sparse" if sparse else "dense"] = {"iterations_per_s": round(args.iters / dt, 1), "views_per_s": round(args.iters * world / dt, 1),
                                                "replicas_identical": bool(torch.equal(lo, hi)), "last_loss": losses[-1]}
    PairPolicy.reset()
    if trace and rank == 0:
        print("iteration ms:", trace, file=sys.stderr, flush=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--single-device", action="store_true")
    ap.add_argument("--n", type=int, default=300_000)
    ap.add_argument("--width", type=int, default=1600)
    ap.add_argument("--height", type=int, default=1200)
    ap.add_argument("--iters", type=int, default=48)
    args = ap.parse_args()
    world, rank, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    di = 0 if args.single_device else local
    torch.cuda.set_device(di)
    dev = torch.device("cuda", di)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(args.backend)
    cdev = dev if args.backend == "nccl" else torch.device("cpu")
    res = run(args, dev, cdev, rank, world)
    if rank == 0:
        print(json.dumps({"workload": f"data-parallel training of one scene: {args.n} Gaussians SH3, {args.width}x{args.height}, one view per rank per iteration, "
                                      f"fused view + FusedAdam, {args.iters} iterations", "ranks": world, "backend": args.backend,
                          "single_device": args.single_device, **res}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
