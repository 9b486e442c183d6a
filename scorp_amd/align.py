"""Pose-hypothesis scoring on the rasterizer — the GPU form of the rotation sweep of config #3.

The reference feeds its `rotations_N.npz` initialisations to Open3D ICP on the CPU
(align_3dgs_clpe_9dof.py:80-111) and keeps the arg-max fitness; rendering is only used afterwards.  Here each
hypothesis is scored by rendering the (SH0) object under the rotation from a handful of cameras, forward only, and
comparing alpha + depth with the cached target renders (the role of the original-scene renders at :336-368).
Hypotheses are independent, so they shard over ranks (scorp_amd.parallel.sweep) with one gather at the end.
"""
import copy

import torch

from .rasterizer3d import PairPolicy
from .renderer import render
from .transforms import gaussians_rotate


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = True


@torch.no_grad()
def render_views(model, cameras, bg):
    return [render(c, model, _Pipe(), bg) for c in cameras]


@torch.no_grad()
def hypothesis_fitness(model, R, cameras, targets, bg):
    """Higher is better: minus the mean |alpha - alpha*| + |depth - depth*| over the cameras."""
    m = copy.copy(model)
    m._xyz, m._rotation = model._xyz.detach().clone(), model._rotation.detach().clone()
    m._features_rest = model._features_rest.detach().clone()
    gaussians_rotate(m, R, fix_center=True)
    err = 0.0
    for cam, tgt in zip(cameras, targets):
        out = render(cam, m, _Pipe(), bg)
        err = err + (out["render_alpha"] - tgt["render_alpha"]).abs().mean() + \
            (out["render_depth"] - tgt["render_depth"]).abs().mean()
    return -(err / len(cameras))


def _sweep_eager(model, rotations, ids, cameras, targets, bg, dev):
    """Renders issued without host synchronisation (`PairPolicy` "reserve"), verified once per hypothesis; a hypothesis
    whose reservation overflowed is re-scored with exact sizing."""
    out = []
    for i in ids:
        R = torch.as_tensor(rotations[i], dtype=torch.float32, device=dev)
        prev = PairPolicy.mode
        PairPolicy.mode = "reserve"
        try:
            f = hypothesis_fitness(model, R, cameras, targets, bg)
            PairPolicy.drain()
        except RuntimeError:
            PairPolicy.mode = "exact"
            f = hypothesis_fitness(model, R, cameras, targets, bg)
        finally:
            PairPolicy.mode = prev
        out.append(f.reshape(1).float())
    return out


def _sweep_graph(model, rotations, ids, cameras, targets, bg, dev):
    """One hypothesis (rotate + len(cameras) renders + comparison) captured as a HIP graph and replayed per rotation:
    the sweep's kernels are tens of microseconds each, so eager launches are CPU-bound (3.1k renders/s eager, 5.1k
    replayed at 100k Gaussians / 800x800).  The pair reservation is learned from an eager pass; every replay's
    overflow flags are folded into a device-side maximum and checked once at the end."""
    prev = PairPolicy.mode
    PairPolicy.mode = "reserve"
    pend_before = PairPolicy._pending
    try:
        PairPolicy._pending = []
        Rbuf = torch.as_tensor(rotations[ids[0]], dtype=torch.float32, device=dev).clone()
        hypothesis_fitness(model, Rbuf, cameras, targets, bg)
        worst = PairPolicy.drain()                       # sizes the reservation (raises if the default was too small)
        PairPolicy.reserve = max(PairPolicy.reserve, int(2.0 * worst) + 4096)   # other rotations see other pair counts
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            hypothesis_fitness(model, Rbuf, cameras, targets, bg)
            PairPolicy.drain()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            fit = hypothesis_fitness(model, Rbuf, cameras, targets, bg)
            states, PairPolicy._pending = PairPolicy._pending, []
            # StateHeader: num_pairs u32 @0, overflow u32 @4
            flags = torch.stack([p.header[:8].view(torch.int32) for p in states]).amax(0)
        worst_flags = torch.zeros(2, dtype=torch.int32, device=dev)
        out = []
        for i in ids:
            Rbuf.copy_(torch.as_tensor(rotations[i], dtype=torch.float32), non_blocking=True)
            graph.replay()
            out.append(fit.reshape(1).float().clone())
            worst_flags = torch.maximum(worst_flags, flags)
        n_pairs, overflow = (int(v) for v in worst_flags.tolist())   # the sweep's one synchronisation
        if overflow:
            PairPolicy.reserve = max(PairPolicy.reserve, int(1.25 * n_pairs) + 4096)
            raise RuntimeError("pair reservation overflowed during graph replay")
        return out
    finally:
        PairPolicy.mode = prev
        PairPolicy._pending = pend_before


def rotation_sweep(model, rotations, cameras, targets, bg, use_graph=None):
    """Score every rotation hypothesis (sharded over ranks); returns (ids, fitness[n,1], best id).  On a GPU the
    hypotheses of this rank are replayed from a captured HIP graph (`use_graph=False`, or any capture failure, falls
    back to eager launches)."""
    from .parallel import gather_results, shard_indices
    dev = model._xyz.device
    mine = shard_indices(len(rotations))
    if use_graph is None:
        use_graph = dev.type == "cuda"
    vals = None
    if use_graph and mine:
        try:
            vals = _sweep_graph(model, rotations, mine, cameras, targets, bg, dev)
        except Exception:   # capture unsupported / reservation overflow: the eager path is always correct
            torch.cuda.synchronize()
            vals = None
    if vals is None:
        vals = _sweep_eager(model, rotations, mine, cameras, targets, bg, dev)
    v = torch.stack(vals) if vals else torch.zeros((0, 1), dtype=torch.float32, device=dev)
    ids, scores = gather_results(mine, v.to(dev))
    best = int(ids[torch.argmax(scores[:, 0])]) if ids.numel() else -1
    return ids, scores, best
