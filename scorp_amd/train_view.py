"""One training view in one library call: `render()` + `0.8 L1 + 0.2 (1 - SSIM)` + `loss.backward()`.

`train_view(camera, pc, pipe, bg, gt_image)` does what lines 88-150 of the reference's train_3dgs.py do for the plain
photometric loss (render -> l1_loss / ssim -> loss.backward()), with the same kernels as `renderer.render` +
`fused_loss.fused_l1_ssim_loss` + autograd, but enqueued by a single C call (`scorp_gs3d_train_view`): at ~1 ms of
device work per view the Python glue, autograd bookkeeping and a dozen ctypes calls cost about as much host time as
the GPU needs, and any hiccup of the host shows up as idle GPU.  No autograd graph is built; the gradients land in the
`.grad` of the model's six raw leaves (accumulating, like backward()), the screen-space gradient in
`out["viewspace_points"].grad` (what add_densification_stats reads, gaussian_model.py:603-605).

The pair buffer is always "reserved" (no host synchronisation): sized by `PairPolicy.capacity()` for this (model size,
resolution, stream), verified by `PairPolicy.drain()`.  The view's overflow word comes back as a device tensor
(`out["overflow"]`): `train()` hands it to `FusedAdam` (the step is skipped on the device if it is set) and masks the
densification statistics with it, so a view that overflowed its reservation changes nothing.
"""
import ctypes
import math

import torch

from . import _C
from .rasterizer3d import GaussianRasterizationSettings, PairPolicy, _backward_flags, _inputs_struct, _prep, _ptr, _stream


def _accumulate(p, g):
    if p.grad is None:
        p.grad = g
    else:
        p.grad += g


def train_view(viewpoint_camera, pc, pipe, bg_color, gt_image, lambda_dssim=0.2, mask=None, scaling_modifier=1.0,
               optimizer=None, stats=None, grad_out=None):
    """Returns the dict of `renderer.render` plus "loss", "l1", "ssim" (0-d views of one device tensor); parameter
    gradients are accumulated into `pc`'s leaves.  Needs the model's raw leaves (fused activations).

    `optimizer` (a FusedAdam holding the model's leaves): the optimizer step of this iteration is applied INSIDE the view, by
    the per-Gaussian backward kernel (ScorpFusedAdam, include/scorp_gs.h) - same update as optimizer.step() on the view's
    gradients, bit for bit, but the gradient rows never go to HBM: the leaves get NO .grad, the result carries
    "optimizer_stepped": True and the caller must not call optimizer.step() for this iteration.  Skipped on the device if
    the view overflowed its pair reservation (counted in optimizer.take_skipped()).  `stats` = (max_radii2D,
    xyz_gradient_accum, denom): the view's share of the densification statistics (GaussianModel.accumulate_view_stats) by
    the same kernel; then "viewspace_points".grad is None.  If the step cannot be fused (pending gradients, a leaf the
    optimizer does not hold) the view runs as without `optimizer` and says "optimizer_stepped": False.

    `grad_out` (six tensors or None entries, the leaves' order xyz, features_dc, features_rest, opacity, scaling, rotation; e.g.
    parallel.GradArena.views): the gradients are WRITTEN there (not accumulated) and become the leaves' .grad - the
    data-parallel loop's collective then reads them where the kernel left them."""
    L = _C.lib()
    xyz = pc.get_xyz
    if not xyz.is_cuda:
        raise RuntimeError("train_view needs GPU tensors (scorp_amd has no CPU path)")
    f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
    leaves = (xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw)
    dev = xyz.device
    w, h = viewpoint_camera.resolution
    W, H, N = int(w), int(h), xyz.shape[0]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False,
        debug=bool(getattr(pipe, "debug", False)))
    t = [_prep(x.detach(), n) for x, n in zip(leaves, ("means3D", "features_dc", "features_rest", "opacity", "scaling", "rotation"))]
    keep = []
    args = _inputs_struct(settings, t[0], t[1], None, t[3], t[4], t[5], None, keep, t[2], 7)
    gt = _prep(gt_image, "gt_image")
    if mask is not None:
        mask = _prep(mask.expand(1, H, W), "mask")
    new = lambda shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=dev)
    color, depth_raw, alpha, depth = new((3, H, W)), new((1, H, W)), new((1, H, W)), new((1, H, W))
    radii, visible = new((N,), torch.int32), new((N,), torch.uint8)
    loss3, grad_color = new((3,)), new((3, H, W))
    state_bytes = L.scorp_gs3d_state_bytes(N, W, H)
    state = new((state_bytes,), torch.uint8)
    capacity = PairPolicy.capacity(N, H, W)
    pairs = new((L.scorp_gs3d_pairs_bytes(capacity),), torch.uint8)
    ws_bytes = L.scorp_loss_workspace_bytes(3, H, W)
    ws = new((ws_bytes,), torch.uint8)
    flags = _backward_flags()     # rasterizer3d.backward_precision(...) / SCORP_BACKWARD_DETERMINISTIC reach the one-call view too
    scratch_bytes = L.scorp_gs3d_backward_scratch_bytes_ex(N, W, H, capacity, flags)
    scratch = new((scratch_bytes,), torch.uint8)
    need = [p.requires_grad for p in leaves]          # frozen leaves (post-refine) get no gradient buffer: NULL = not wanted
    need[1] = need[2] = need[1] or need[2]            # the SH gradient is written as a whole
    pack = None
    if (optimizer is not None and hasattr(optimizer, "fused_view_pack") and f_rest.numel() > 0
            and all(x.data_ptr() == p.data_ptr() for x, p in zip(t, leaves))):
        st = None
        if stats is not None and xyz.requires_grad:
            st = tuple(s_ if (s_.dtype == torch.float32 and s_.is_contiguous()) else None for s_ in stats)
            st = None if any(s_ is None for s_ in st) else st
        pack = optimizer.fused_view_pack(leaves, st)
        stats = st
    fused_step = pack is not None
    g = [torch.empty_like(x) if (n and not fused_step) else None for x, n in zip(t, need)]
    if grad_out is not None and not fused_step:
        for k, (x, n, go) in enumerate(zip(t, need, grad_out)):
            if n and go is not None:
                assert go.shape == x.shape and go.dtype == torch.float32 and go.is_contiguous() and go.device == x.device
                g[k] = go
    # the screen-space gradient feeds the densification statistics: not produced when the positions are frozen
    # (renderer.render does the same), which leaves the backward with colour gradients only -> its colour-only path
    g_means2D = new((N, 3)) if (xyz.requires_grad and not (fused_step and stats is not None)) else None
    grads = _C.ScorpGs3dGrads()
    grads.means3D, grads.means2D, grads.shs, grads.shs_rest = _ptr(g[0]), _ptr(g_means2D), _ptr(g[1]), _ptr(g[2])
    grads.opacities, grads.scales, grads.rotations = _ptr(g[3]), _ptr(g[4]), _ptr(g[5])
    v = _C.ScorpGs3dTrainView()
    v.inputs = ctypes.addressof(args)
    v.out_radii, v.state, v.state_bytes, v.pairs, v.capacity = radii.data_ptr(), state.data_ptr(), state_bytes, pairs.data_ptr(), capacity
    v.out_color, v.out_depth_raw, v.out_alpha = color.data_ptr(), depth_raw.data_ptr(), alpha.data_ptr()
    v.out_depth, v.out_visible = depth.data_ptr(), visible.data_ptr()
    v.gt, v.mask, v.lambda_dssim = gt.data_ptr(), (None if mask is None else mask.data_ptr()), float(lambda_dssim)
    v.backward_flags = flags
    v.out_loss3, v.loss_workspace, v.loss_workspace_bytes = loss3.data_ptr(), ws.data_ptr(), ws_bytes
    v.grad_color, v.grads = grad_color.data_ptr(), ctypes.addressof(grads)
    v.backward_scratch, v.backward_scratch_bytes = scratch.data_ptr(), scratch_bytes
    header = new((64,), torch.uint8)                  # {pairs needed, overflow, capacity, 0}: written by the scatter kernel
    v.out_header = header.data_ptr()
    if fused_step:
        v.adam = ctypes.addressof(pack[0])
    _C.check(L.scorp_gs3d_train_view(ctypes.byref(v), _stream()), "scorp_gs3d_train_view")
    PairPolicy.pend(state, N, H, W, header=header)    # queued for drain(): no copy launch, the state blob is not pinned
    if not fused_step:
        for k, (p, gp) in enumerate(zip(leaves, g)):
            if p.requires_grad:
                if grad_out is not None and grad_out[k] is not None:
                    p.grad = gp.view_as(p)      # written in place of whatever was there: the arena is the gradient
                else:
                    _accumulate(p, gp.view_as(p))
    return {"optimizer_stepped": fused_step, "stats_accumulated": fused_step and stats is not None, "render": color, "viewspace_points": _ViewspaceGrad(g_means2D), "visibility_filter": visible.view(torch.bool),
            "radii": radii, "render_depth": depth, "render_alpha": alpha, "loss": loss3[0], "l1": loss3[1], "ssim": loss3[2],
            # != 0 if this view needed more pairs than were reserved (its images and gradients then come from truncated
            # tile lists): a device word, so the caller can make the optimizer step conditional without a host sync
            "overflow": header.view(torch.int32)[1:2]}


class _ViewspaceGrad:
    """What the training loop uses of `viewspace_points`: its `.grad` ([N,3], gaussian_model.py:603-605)."""
    __slots__ = ("grad",)

    def __init__(self, grad):
        self.grad = grad


def train_view2d(viewpoint_camera, pc, pipe, bg_color, gt_image, lambda_dssim=0.2, lambda_normal=0.0, lambda_dist=0.0,
                 mask=None, scaling_modifier=1.0, optimizer=None, stats=None):
    """The 2DGS twin of `train_view`: one iteration of train_2dgs.py:95-150 for the plain photometric loss plus the
    normal-consistency / depth-distortion regularisers (train_2dgs.py:142-150), enqueued by ONE library call
    (`scorp_gs2d_train_view`).  Returns "render", "allmap", "radii", "visibility_filter", "viewspace_points",
    "loss" (= photometric + normal + distortion, a 0-d device tensor), "l1", "ssim", "normal_loss", "dist_loss",
    "overflow"; parameter gradients are accumulated into the surfel model's leaves.  `optimizer` / `stats`: the optimizer
    step and the view's densification statistics inside the view, as for `train_view` (ScorpGs2dTrainView.adam)."""
    from .renderer2d import _camera_rays
    L = _C.lib()
    xyz = pc.get_xyz
    if not xyz.is_cuda:
        raise RuntimeError("train_view2d needs GPU tensors (scorp_amd has no CPU path)")
    f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
    leaves = (xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw)
    dev = xyz.device
    w, h = viewpoint_camera.resolution
    W, H, N = int(w), int(h), xyz.shape[0]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=math.tan(viewpoint_camera.FoVx * 0.5),
        tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color, scale_modifier=scaling_modifier,
        viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
        sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
    t = [_prep(x.detach(), n) for x, n in zip(leaves, ("means3D", "features_dc", "features_rest", "opacity", "scaling", "rotation"))]
    keep = []
    args = _inputs_struct(settings, t[0], t[1], None, t[3], t[4], t[5], None, keep, t[2], 7)
    gt = _prep(gt_image, "gt_image")
    if mask is not None:
        mask = _prep(mask.expand(1, H, W), "mask")
    new = lambda shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=dev)
    color, allmap = new((3, H, W)), new((7, H, W))
    radii = new((N,), torch.int32)
    loss5, grad_color, grad_allmap = new((5,)), new((3, H, W)), new((7, H, W))
    state_bytes = L.scorp_gs2d_state_bytes(N, W, H)
    state = new((state_bytes,), torch.uint8)
    capacity = PairPolicy.capacity(N, H, W)
    pairs = new((L.scorp_gs3d_pairs_bytes(capacity),), torch.uint8)
    ws_bytes = L.scorp_loss_workspace_bytes(3, H, W)
    ws = new((ws_bytes,), torch.uint8)
    rws_bytes = L.scorp_gs2d_regularizers_workspace_bytes(W, H)
    rws = new((rws_bytes,), torch.uint8)
    flags = _backward_flags() & ~_C.BACKWARD_SCRATCH_ZEROED   # backward_precision(...) reaches the 2DGS one-call view too
    scratch_bytes = L.scorp_gs2d_backward_scratch_bytes_ex(N, W, H, capacity, flags)
    scratch = new((scratch_bytes,), torch.uint8)
    need = [p.requires_grad for p in leaves]
    need[1] = need[2] = need[1] or need[2]
    pack = None
    if (optimizer is not None and hasattr(optimizer, "fused_view_pack") and f_rest.numel() > 0
            and all(x.data_ptr() == p.data_ptr() for x, p in zip(t, leaves))):
        st = None
        if stats is not None and xyz.requires_grad:
            st = tuple(s_ if (s_.dtype == torch.float32 and s_.is_contiguous()) else None for s_ in stats)
            st = None if any(s_ is None for s_ in st) else st
        pack = optimizer.fused_view_pack(leaves, st)
        stats = st
    fused_step = pack is not None
    g = [torch.empty_like(x) if (n and not fused_step) else None for x, n in zip(t, need)]
    g_means2D = new((N, 3)) if (xyz.requires_grad and not (fused_step and stats is not None)) else None
    grads = _C.ScorpGs3dGrads()
    grads.means3D, grads.means2D, grads.shs, grads.shs_rest = _ptr(g[0]), _ptr(g_means2D), _ptr(g[1]), _ptr(g[2])
    grads.opacities, grads.scales, grads.rotations = _ptr(g[3]), _ptr(g[4]), _ptr(g[5])
    rays_d, rays_o = _camera_rays(viewpoint_camera, dev)
    v = _C.ScorpGs2dTrainView()
    v.inputs = ctypes.addressof(args)
    v.out_radii, v.state, v.state_bytes, v.pairs, v.capacity = radii.data_ptr(), state.data_ptr(), state_bytes, pairs.data_ptr(), capacity
    v.out_color, v.out_allmap = color.data_ptr(), allmap.data_ptr()
    v.gt, v.mask = gt.data_ptr(), (None if mask is None else mask.data_ptr())
    v.rays_d, v.rays_o = rays_d.data_ptr(), rays_o.data_ptr()
    v.lambda_dssim, v.depth_ratio = float(lambda_dssim), float(getattr(pipe, "depth_ratio", 1.0))
    v.lambda_normal, v.lambda_dist = float(lambda_normal), float(lambda_dist)
    v.out_loss3, v.out_reg2 = loss5.data_ptr(), loss5[3:].data_ptr()
    v.loss_workspace, v.loss_workspace_bytes, v.reg_workspace, v.reg_workspace_bytes = ws.data_ptr(), ws_bytes, rws.data_ptr(), rws_bytes
    v.grad_color, v.grad_allmap, v.grads = grad_color.data_ptr(), grad_allmap.data_ptr(), ctypes.addressof(grads)
    v.backward_scratch, v.backward_scratch_bytes = scratch.data_ptr(), scratch_bytes
    v.backward_flags = flags
    if fused_step:
        v.adam = ctypes.addressof(pack[0])
    _C.check(L.scorp_gs2d_train_view(ctypes.byref(v), _stream()), "scorp_gs2d_train_view")
    header = PairPolicy.pend(state, N, H, W)
    if not fused_step:
        for p, gp in zip(leaves, g):
            if p.requires_grad:
                _accumulate(p, gp.view_as(p))
    return {"optimizer_stepped": fused_step, "stats_accumulated": fused_step and stats is not None, "render": color, "allmap": allmap, "viewspace_points": _ViewspaceGrad(g_means2D), "visibility_filter": radii > 0,
            "radii": radii, "loss": loss5[0] + loss5[3] + loss5[4], "l1": loss5[1], "ssim": loss5[2],
            "normal_loss": loss5[3], "dist_loss": loss5[4], "overflow": header.view(torch.int32)[1:2]}
