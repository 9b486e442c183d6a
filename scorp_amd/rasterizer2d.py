"""`diff_surfel_rasterization`-compatible front-end (2DGS) over libscorp_gs.so.

Same names / call signature as the module imported at gs2dgs/gaussian_renderer/__init__.py:14 and called at :51-67,
:111-120: `GaussianRasterizer(raster_settings)(means3D, means2D, shs, colors_precomp, opacities, scales[N,2],
rotations, cov3D_precomp[N,9]) -> (color[3,H,W], radii[N], allmap[7,H,W])`.
"""
import ctypes

import torch
import torch.nn as nn

from . import _C
from . import rasterizer3d as R3
from .rasterizer3d import (GaussianRasterizationSettings, LAST_NUM_PAIRS_LOG, PairPolicy, _inputs_struct, _prep, _ptr,
                           _stream)


def _forward2d(ctx, settings, means3D, sh, sh_rest, colors_precomp, opacities, scales, rotations, transmat, raw):
    L = _C.lib()
    dev = means3D.device
    N, H, W = means3D.shape[0], int(settings.image_height), int(settings.image_width)
    keep = []
    args = _inputs_struct(settings, means3D, sh, colors_precomp, opacities, scales, rotations, transmat, keep, sh_rest, raw)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    allmap = torch.empty((7, H, W), dtype=torch.float32, device=dev)
    radii = torch.empty((N,), dtype=torch.int32, device=dev)
    sb = L.scorp_gs2d_state_bytes(N, W, H)
    state = torch.empty(sb, dtype=torch.uint8, device=dev)
    stream = _stream()
    _C.check(L.scorp_gs2d_preprocess(ctypes.byref(args), _ptr(radii), _ptr(state), sb, stream), "scorp_gs2d_preprocess")
    if PairPolicy.mode == "exact":
        n = ctypes.c_uint64(0)
        _C.check(L.scorp_gs3d_num_pairs(_ptr(state), stream, ctypes.byref(n)), "scorp_gs3d_num_pairs")
        capacity = max(int(n.value), 1)
        LAST_NUM_PAIRS_LOG.append(int(n.value))
        del LAST_NUM_PAIRS_LOG[:-64]
    else:
        capacity = PairPolicy.capacity(N, H, W)
    pairs = torch.empty(L.scorp_gs3d_pairs_bytes(capacity), dtype=torch.uint8, device=dev)
    fn = L.scorp_gs2d_render if R3.want_backward(ctx) else L.scorp_gs2d_render_image   # nothing to differentiate
    _C.check(fn(ctypes.byref(args), _ptr(state), _ptr(pairs), capacity, _ptr(color), _ptr(allmap), stream), "scorp_gs2d_render")
    if PairPolicy.mode != "exact":
        PairPolicy.pend(state, N, H, W)   # the StateHeader only (see rasterizer3d.PairPolicy.pend)
    ctx.settings, ctx.capacity = settings, capacity
    # rasterizer3d.backward_precision("exact_fp32" / "deterministic" / ...) and SCORP_BACKWARD_DETERMINISTIC reach the surfel
    # rasterizer too (scorp_gs2d_backward_ex); the flag is fixed when the forward is issued, as in 3-D
    ctx.backward_flags = R3._backward_flags() & ~_C.BACKWARD_SCRATCH_ZEROED
    return color, radii, allmap, state, pairs, keep


def _backward2d(ctx, args, N, dev, state, pairs, grad_color, grad_allmap, grads):
    L = _C.lib()
    gc = _prep(grad_color, "grad_color")
    ga = _prep(grad_allmap, "grad_allmap") if grad_allmap is not None else None
    s = ctx.settings
    sbytes = L.scorp_gs2d_backward_scratch_bytes_ex(N, int(s.image_width), int(s.image_height), ctx.capacity, ctx.backward_flags)
    scratch = torch.empty(sbytes, dtype=torch.uint8, device=dev)
    _C.check(L.scorp_gs2d_backward_ex(ctypes.byref(args), _ptr(state), _ptr(pairs), ctx.capacity, _ptr(gc), _ptr(ga),
                                      ctypes.byref(grads), _ptr(scratch), sbytes, ctx.backward_flags, _stream()),
             "scorp_gs2d_backward")


class _RasterizeSurfels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, transmat, settings):
        dev = means3D.device
        means3D, sh, colors_precomp = _prep(means3D, "means3D"), _prep(sh, "shs"), _prep(colors_precomp, "colors_precomp")
        opacities, scales, rotations, transmat = (_prep(opacities, "opacities"), _prep(scales, "scales"),
                                                  _prep(rotations, "rotations"), _prep(transmat, "cov3D_precomp"))
        color, radii, allmap, state, pairs, keep = _forward2d(ctx, settings, means3D, sh, None, colors_precomp, opacities,
                                                              scales, rotations, transmat, raw=0)
        ctx.has = (sh is not None, colors_precomp is not None, scales is not None, transmat is not None)
        none = torch.empty(0, device=dev)
        ctx.save_for_backward(means3D, none if sh is None else sh, none if colors_precomp is None else colors_precomp,
                              opacities, none if scales is None else scales, none if rotations is None else rotations,
                              none if transmat is None else transmat, state, pairs, *keep)
        ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_allmap):
        means3D, sh, colors_precomp, opacities, scales, rotations, transmat, state, pairs, bg, vm, pm, cp = ctx.saved_tensors
        has_sh, has_col, has_sr, has_tm = ctx.has
        sh = sh if has_sh else None
        colors_precomp = colors_precomp if has_col else None
        scales, rotations = (scales, rotations) if has_sr else (None, None)
        transmat = transmat if has_tm else None
        s = ctx.settings._replace(bg=bg, viewmatrix=vm, projmatrix=pm, campos=cp)
        keep = []
        args = _inputs_struct(s, means3D, sh, colors_precomp, opacities, scales, rotations, transmat, keep)
        N, dev = means3D.shape[0], means3D.device
        need = ctx.needs_input_grad
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_means3D = new(N, 3) if need[0] else None
        g_means2D = new(N, 3) if need[1] else None
        g_sh = torch.empty_like(sh) if (need[2] and has_sh) else None
        g_col = new(N, 3) if (need[3] and has_col) else None
        g_op = torch.empty_like(opacities) if need[4] else None
        g_sc = new(N, 2) if (need[5] and has_sr) else None
        g_rot = new(N, 4) if (need[6] and has_sr) else None
        g_tm = new(N, 9) if (need[7] and has_tm) else None
        grads = _C.ScorpGs3dGrads()
        grads.means3D, grads.means2D, grads.shs, grads.colors_precomp = _ptr(g_means3D), _ptr(g_means2D), _ptr(g_sh), _ptr(g_col)
        grads.opacities, grads.scales, grads.rotations, grads.cov3D_precomp = _ptr(g_op), _ptr(g_sc), _ptr(g_rot), _ptr(g_tm)
        _backward2d(ctx, args, N, dev, state, pairs, grad_color, grad_allmap, grads)
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_tm, None


class _RasterizeSurfelsRaw(torch.autograd.Function):
    """Raw-leaf variant (logit opacity, log scale[N,2], un-normalised quaternion, dc/rest split): see rasterizer3d."""

    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, settings):
        means3D, f_dc, f_rest = _prep(means3D, "means3D"), _prep(f_dc, "features_dc"), _prep(f_rest, "features_rest")
        opacity_raw, scaling_raw, rotation_raw = _prep(opacity_raw, "opacity"), _prep(scaling_raw, "scaling"), _prep(rotation_raw, "rotation")
        color, radii, allmap, state, pairs, keep = _forward2d(ctx, settings, means3D, f_dc, f_rest, None, opacity_raw,
                                                              scaling_raw, rotation_raw, None, raw=7)
        ctx.save_for_backward(means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, state, pairs, *keep)
        ctx.mark_non_differentiable(radii)
        return color, radii, allmap

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_allmap):
        means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, state, pairs, bg, vm, pm, cp = ctx.saved_tensors
        s = ctx.settings._replace(bg=bg, viewmatrix=vm, projmatrix=pm, campos=cp)
        keep = []
        args = _inputs_struct(s, means3D, f_dc, None, opacity_raw, scaling_raw, rotation_raw, None, keep, f_rest, 7)
        N, dev = means3D.shape[0], means3D.device
        need = ctx.needs_input_grad
        new = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        g_means3D = new(N, 3) if need[0] else None
        g_means2D = new(N, 3) if need[1] else None
        want_sh = need[2] or need[3]
        g_dc = torch.empty_like(f_dc) if want_sh else None
        g_rest = torch.empty_like(f_rest) if want_sh else None
        g_op = torch.empty_like(opacity_raw) if need[4] else None
        g_sc = new(N, 2) if need[5] else None
        g_rot = new(N, 4) if need[6] else None
        grads = _C.ScorpGs3dGrads()
        grads.means3D, grads.means2D, grads.shs, grads.shs_rest = _ptr(g_means3D), _ptr(g_means2D), _ptr(g_dc), _ptr(g_rest)
        grads.opacities, grads.scales, grads.rotations = _ptr(g_op), _ptr(g_sc), _ptr(g_rot)
        _backward2d(ctx, args, N, dev, state, pairs, grad_color, grad_allmap, grads)
        return g_means3D, g_means2D, g_dc if need[2] else None, g_rest if need[3] else None, g_op, g_sc, g_rot, None


def rasterize_surfels_raw(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    R3._tls.grad_mode = torch.is_grad_enabled()
    return _RasterizeSurfelsRaw.apply(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            vm = self.raster_settings.viewmatrix
            return (positions @ vm[:3, 2] + vm[3, 2]) > 0.2

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        R3._tls.grad_mode = torch.is_grad_enabled()
        return _RasterizeSurfels.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                       self.raster_settings)


class _SurfelMaps(torch.autograd.Function):
    """allmap -> (render_alpha, render_normal, render_dist, surf_depth, surf_normal): the per-pixel tail of
    gs2dgs/gaussian_renderer/__init__.py:131-160 as one HIP kernel each way (scorp_gs2d_maps_forward/backward)."""

    @staticmethod
    def forward(ctx, allmap, viewmatrix, rays_d, rays_o, depth_ratio):
        L = _C.lib()
        allmap = _prep(allmap, "allmap")
        _, H, W = allmap.shape
        dev = allmap.device
        out = torch.empty((9, H, W), dtype=torch.float32, device=dev)   # alpha | normal(3) | dist | depth | surf_normal(3)
        ra, rn, rd, sd, sn = out[0:1], out[1:4], out[4:5], out[5:6], out[6:9]
        _C.check(L.scorp_gs2d_maps_forward(W, H, _ptr(allmap), _ptr(viewmatrix), _ptr(rays_d), _ptr(rays_o),
                                           float(depth_ratio), _ptr(ra), _ptr(rn), _ptr(rd), _ptr(sd), _ptr(sn), _stream()),
                 "scorp_gs2d_maps_forward")
        ctx.save_for_backward(allmap, viewmatrix, rays_d, rays_o, sd)
        ctx.depth_ratio = float(depth_ratio)
        ctx.set_materialize_grads(False)
        return ra, rn, rd, sd, sn

    @staticmethod
    def backward(ctx, g_ra, g_rn, g_rd, g_sd, g_sn):
        L = _C.lib()
        allmap, viewmatrix, rays_d, rays_o, sd = ctx.saved_tensors
        _, H, W = allmap.shape
        gs = [None if g is None else _prep(g, "grad") for g in (g_ra, g_rn, g_rd, g_sd, g_sn)]
        g_allmap = torch.empty_like(allmap)
        _C.check(L.scorp_gs2d_maps_backward(W, H, _ptr(allmap), _ptr(viewmatrix), _ptr(rays_d), _ptr(rays_o), ctx.depth_ratio,
                                            _ptr(sd), _ptr(gs[0]), _ptr(gs[1]), _ptr(gs[2]), _ptr(gs[3]), _ptr(gs[4]),
                                            _ptr(g_allmap), _stream()), "scorp_gs2d_maps_backward")
        return g_allmap, None, None, None, None


def surfel_maps(allmap, viewmatrix, rays_d, rays_o, depth_ratio):
    """(render_alpha[1,H,W], render_normal[3,H,W], render_dist[1,H,W], surf_depth[1,H,W], surf_normal[3,H,W])."""
    if not allmap.is_cuda:
        raise RuntimeError("surfel_maps needs CUDA/HIP tensors: scorp_amd has no CPU fallback")
    return _SurfelMaps.apply(allmap, _prep(viewmatrix, "viewmatrix"), _prep(rays_d, "rays_d"), _prep(rays_o, "rays_o"),
                             depth_ratio)


class _SurfelRegularizers(torch.autograd.Function):
    """(normal_loss, dist_loss) of train_2dgs.py:142-150 straight from allmap: scorp_gs2d_regularizers_forward/backward."""

    @staticmethod
    def forward(ctx, allmap, viewmatrix, rays_d, rays_o, depth_ratio, lambda_normal, lambda_dist):
        L = _C.lib()
        allmap = _prep(allmap, "allmap")
        _, H, W = allmap.shape
        out = torch.empty(2, dtype=torch.float32, device=allmap.device)
        wb = L.scorp_gs2d_regularizers_workspace_bytes(W, H)
        ws = torch.empty(wb, dtype=torch.uint8, device=allmap.device)
        _C.check(L.scorp_gs2d_regularizers_forward(W, H, _ptr(allmap), _ptr(viewmatrix), _ptr(rays_d), _ptr(rays_o),
                                                   float(depth_ratio), float(lambda_normal), float(lambda_dist), _ptr(out),
                                                   _ptr(ws), wb, _stream()), "scorp_gs2d_regularizers_forward")
        ctx.save_for_backward(allmap, viewmatrix, rays_d, rays_o)
        ctx.consts = (float(depth_ratio), float(lambda_normal), float(lambda_dist))
        return out

    @staticmethod
    def backward(ctx, g_out):
        L = _C.lib()
        allmap, viewmatrix, rays_d, rays_o = ctx.saved_tensors
        _, H, W = allmap.shape
        g_out = _prep(g_out, "grad")
        g_allmap = torch.empty_like(allmap)
        dr, ln, ld = ctx.consts
        _C.check(L.scorp_gs2d_regularizers_backward(W, H, _ptr(allmap), _ptr(viewmatrix), _ptr(rays_d), _ptr(rays_o), dr, ln, ld,
                                                    _ptr(g_out), _ptr(g_allmap), _stream()), "scorp_gs2d_regularizers_backward")
        return g_allmap, None, None, None, None, None, None


def surfel_regularizer_losses(allmap, viewmatrix, rays_d, rays_o, depth_ratio, lambda_normal, lambda_dist):
    """tensor[2] = (lambda_normal * mean(1 - render_normal . surf_normal), lambda_dist * mean(render_dist))."""
    if not allmap.is_cuda:
        raise RuntimeError("surfel_regularizer_losses needs CUDA/HIP tensors: scorp_amd has no CPU fallback")
    return _SurfelRegularizers.apply(allmap, _prep(viewmatrix, "viewmatrix"), _prep(rays_d, "rays_d"), _prep(rays_o, "rays_o"),
                                     depth_ratio, lambda_normal, lambda_dist)
