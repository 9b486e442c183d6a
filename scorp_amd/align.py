"""Pose-hypothesis scoring on the rasterizer — the GPU form of the rotation sweep of config #3.

The reference feeds its `rotations_N.npz` initialisations to Open3D ICP on the CPU
(align_3dgs_clpe_9dof.py:80-111) and keeps the arg-max fitness; rendering is only used afterwards.  Here each
hypothesis is scored by rendering the (SH0) object under the rotation from a handful of cameras, forward only, and
comparing alpha + depth with the cached target renders (the role of the original-scene renders at :336-368).
Hypotheses are independent, so they shard over ranks (scorp_amd.parallel.sweep) with one gather at the end.
"""
import copy

import torch

from .parallel import sweep
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


def rotation_sweep(model, rotations, cameras, targets, bg):
    """Score every rotation hypothesis (sharded over ranks); returns (ids, fitness[n,1], best id)."""
    dev = model._xyz.device
    return sweep(len(rotations), lambda i: hypothesis_fitness(model, torch.as_tensor(rotations[i], dtype=torch.float32, device=dev),
                                                              cameras, targets, bg), device=dev)
