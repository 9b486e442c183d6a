"""V views of ONE Gaussian set in one launch set (`ScorpGs3dInputs.num_views`, include/scorp_gs.h), forward only.

The align loop renders an object from ~15 cameras per pose hypothesis (align_3dgs_clpe_9dof.py:157-169,336-368).  A
100 k-Gaussian object at 800x800 cannot fill an MI355X: one such render is a chain of seven launch-latency-bound kernels
(147 us for ~25 MB of traffic).  Stacked, the V views are ONE image of V x H rows and V x N virtual Gaussians - the size
of the training benchmark's frame - and go through the same preprocess -> bin -> sort -> blend kernels once.
"""
import ctypes
import math

import torch

from . import _C
from .rasterizer3d import GaussianRasterizationSettings, PairPolicy, _inputs_struct, _prep, _ptr, _stream


class ViewStack:
    """The cameras of a stacked render: matrices as [V,4,4] / [V,3] device tensors (row-vector convention, as
    `Camera.world_view_transform` / `full_proj_transform` / `camera_center`), one resolution and field of view."""

    def __init__(self, cameras, device=None):
        c0 = cameras[0]
        for c in cameras:
            if tuple(c.resolution) != tuple(c0.resolution) or abs(c.FoVx - c0.FoVx) > 1e-12 or abs(c.FoVy - c0.FoVy) > 1e-12:
                raise ValueError("a ViewStack needs cameras of one resolution and field of view")
        dev = device or c0.world_view_transform.device
        self.W, self.H = int(c0.resolution[0]), int(c0.resolution[1])
        if self.H % 16:
            raise ValueError("stacked views need an image height that is a multiple of 16")
        self.tanfovx, self.tanfovy = math.tan(c0.FoVx * 0.5), math.tan(c0.FoVy * 0.5)
        self.view = torch.stack([c.world_view_transform.to(dev) for c in cameras]).float().contiguous()
        self.proj = torch.stack([c.full_proj_transform.to(dev) for c in cameras]).float().contiguous()
        self.campos = torch.stack([c.camera_center.to(dev) for c in cameras]).float().contiguous()
        self.V = len(cameras)

    def moved(self, R, d):
        """The same cameras as seen from an object that was moved by x -> R x + d (R [...,3,3], d [...,3]; leading
        hypothesis dimensions broadcast): rendering the MOVED object from these cameras equals rendering the object as
        it is from the returned ones.  In the row-vector convention x_row -> x_row R^T + d, so every matrix is
        multiplied from the left by M = [[R^T, 0], [d, 1]]; the camera centre goes to (c - d) R.  Returns
        (view [...,V,4,4], proj [...,V,4,4], campos [...,V,3])."""
        R, d = R.to(self.view), d.to(self.view)
        lead = R.shape[:-2]
        M = torch.zeros(lead + (4, 4), dtype=self.view.dtype, device=self.view.device)
        M[..., :3, :3] = R.transpose(-1, -2)
        M[..., 3, :3] = d
        M[..., 3, 3] = 1.0
        view = M.unsqueeze(-3) @ self.view
        proj = M.unsqueeze(-3) @ self.proj
        campos = (self.campos - d.unsqueeze(-2)) @ R
        return view.contiguous(), proj.contiguous(), campos.contiguous()


@torch.no_grad()
def render_stacked(pc, stack, bg, view=None, proj=None, campos=None, scaling_modifier=1.0):
    """Forward-only render of `pc` from the V cameras of `stack` (optionally with replaced matrices, e.g. from
    `stack.moved`).  Returns {"render": [3, V*H, W], "render_depth_raw": [V*H, W] (sum z alpha T, not normalised),
    "render_alpha": [V*H, W], "radii": [V, N], "num_pairs": the (tile, splat) pair count in PairPolicy's "exact" mode, else
    None}; view v is rows [v*H, (v+1)*H)."""
    L = _C.lib()
    xyz = pc.get_xyz
    if not xyz.is_cuda:
        raise RuntimeError("render_stacked needs GPU tensors (scorp_amd has no CPU path)")
    dev = xyz.device
    V, W, H, N = stack.V, stack.W, stack.H, xyz.shape[0]
    view = stack.view if view is None else view
    proj = stack.proj if proj is None else proj
    campos = stack.campos if campos is None else campos
    f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
    t = [_prep(x.detach(), n) for x, n in zip((xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw),
                                              ("means3D", "features_dc", "features_rest", "opacity", "scaling", "rotation"))]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=stack.tanfovx, tanfovy=stack.tanfovy, bg=bg, scale_modifier=scaling_modifier,
        viewmatrix=view, projmatrix=proj, sh_degree=pc.active_sh_degree, campos=campos, prefiltered=False, debug=False)
    keep = []
    args = _inputs_struct(settings, t[0], t[1], None, t[3], t[4], t[5], None, keep, t[2], 7)
    args.num_views = V
    Nt, Ht = V * N, V * H
    new = lambda shape, dtype=torch.float32: torch.empty(shape, dtype=dtype, device=dev)
    color, depth, alpha, radii = new((3, Ht, W)), new((Ht, W)), new((Ht, W)), new((Nt,), torch.int32)
    state_bytes = L.scorp_gs3d_state_bytes(Nt, W, Ht)
    state = new((state_bytes,), torch.uint8)
    stream = _stream()
    _C.check(L.scorp_gs3d_preprocess(ctypes.byref(args), _ptr(radii), _ptr(state), state_bytes, stream), "scorp_gs3d_preprocess")
    if PairPolicy.mode == "exact":
        n = ctypes.c_uint64(0)
        _C.check(L.scorp_gs3d_num_pairs(_ptr(state), stream, ctypes.byref(n)), "scorp_gs3d_num_pairs")
        capacity = max(int(n.value), 1)
        num_pairs = int(n.value)
    else:
        capacity = PairPolicy.capacity(Nt, Ht, W)
        num_pairs = None
    pairs = new((L.scorp_gs3d_pairs_bytes(capacity),), torch.uint8)
    _C.check(L.scorp_gs3d_render_image(ctypes.byref(args), _ptr(state), _ptr(pairs), capacity, _ptr(color), _ptr(depth),
                                       _ptr(alpha), stream), "scorp_gs3d_render_image")
    if PairPolicy.mode != "exact":
        PairPolicy.pend(state, Nt, Ht, W)
    return {"render": color, "render_depth_raw": depth, "render_alpha": alpha, "radii": radii.view(V, N), "num_pairs": num_pairs}


@torch.no_grad()
def score_stacked(pc, stack, bg, view, proj, campos, tgt_depth, tgt_alpha, acc, rows_per_score, scale, scaling_modifier=1.0):
    """`render_stacked` for a caller that only wants the mismatch with a target (the rotation sweep): preprocess + bin + sort
    + the scoring form of the blend (scorp_gs3d_render_score) - no images, no colour, depth and alpha compared with
    tgt_depth / tgt_alpha [rows_per_score, W] where they are formed.  acc[j] += scale * (mismatch of rows
    [j, j + 1) * rows_per_score of the stacked image); `acc`: float32 device tensor with V * H / rows_per_score entries.
    Reserve-mode sizing only (the sweep verifies its launch sets with one PairPolicy.drain())."""
    L = _C.lib()
    xyz = pc.get_xyz
    dev = xyz.device
    V, W, H, N = stack.V, stack.W, stack.H, xyz.shape[0]
    f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
    t = [_prep(x.detach(), n) for x, n in zip((xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw),
                                              ("means3D", "features_dc", "features_rest", "opacity", "scaling", "rotation"))]
    settings = GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=stack.tanfovx, tanfovy=stack.tanfovy, bg=bg, scale_modifier=scaling_modifier,
        viewmatrix=view, projmatrix=proj, sh_degree=pc.active_sh_degree, campos=campos, prefiltered=False, debug=False)
    keep = []
    args = _inputs_struct(settings, t[0], t[1], None, t[3], t[4], t[5], None, keep, t[2], 7)
    args.num_views = V
    Nt, Ht = V * N, V * H
    assert acc.dtype == torch.float32 and acc.is_contiguous() and acc.numel() * rows_per_score == Ht
    assert tgt_depth.is_contiguous() and tgt_alpha.is_contiguous() and tgt_depth.numel() == rows_per_score * W == tgt_alpha.numel()
    radii = torch.empty((Nt,), dtype=torch.int32, device=dev)
    state_bytes = L.scorp_gs3d_state_bytes(Nt, W, Ht)
    state = torch.empty((state_bytes,), dtype=torch.uint8, device=dev)
    stream = _stream()
    _C.check(L.scorp_gs3d_preprocess(ctypes.byref(args), _ptr(radii), _ptr(state), state_bytes, stream), "scorp_gs3d_preprocess")
    capacity = PairPolicy.capacity(Nt, Ht, W)
    pairs = torch.empty((L.scorp_gs3d_pairs_bytes(capacity),), dtype=torch.uint8, device=dev)
    _C.check(L.scorp_gs3d_render_score(ctypes.byref(args), _ptr(state), _ptr(pairs), capacity, _ptr(tgt_depth), _ptr(tgt_alpha),
                                       int(rows_per_score), float(scale), _ptr(acc), stream), "scorp_gs3d_render_score")
    PairPolicy.pend(state, Nt, Ht, W)
