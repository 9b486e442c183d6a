"""`render()` of the 3DGS path — host-side mirror of gs3dgs/gaussian_renderer/__init__.py:24-132.

Same signature, same branches (`pipe.compute_cov3D_python`, `pipe.convert_SHs_python`, `override_color`), same
six-key result dict, same depth normalisation (`render_depth = depth / alpha`, NaN -> 0).  The only deliberate
difference: tensors are created on `pc.get_xyz.device` instead of the literal "cuda".
"""
import math

import torch

from .rasterizer3d import GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians_raw, render_tail
from .sh import eval_sh


_ZEROS = {}



def _fused_activations(pipe, pc):
    """`pipe.fused_activations` if the caller set it; otherwise yes for this package's GaussianModel with its stock
    activations (gaussian_model.GaussianModel.stock_activations) - same numbers, no torch activations / cat per view."""
    flag = getattr(pipe, "fused_activations", None)
    if flag is None:
        return hasattr(pc, "raw_leaves") and getattr(pc, "stock_activations", lambda: False)()
    return bool(flag) and hasattr(pc, "raw_leaves")

def _grad_sink(xyz, requires_grad=True):
    """A fresh leaf of zeros shaped like xyz (the screen-space gradient sink) over a cached, never-written storage."""
    key = (xyz.shape[0], xyz.dtype, xyz.device)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros_like(xyz, requires_grad=False)
    return z.detach().requires_grad_(True) if requires_grad else z


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    xyz = pc.get_xyz
    # gradient sink for the screen-space means; the reference builds it as `zeros_like(...) + 0` with retain_grad()
    # (gaussian_renderer/__init__.py:39-43) — a leaf with requires_grad gives the caller the same `.grad` without the
    # extra 12 MB add kernel per view; the zeros themselves are shared between views (nothing ever writes them), so
    # not even a fill kernel runs: every view gets a fresh leaf over the same storage
    # ... unless the positions are frozen (post_refine_gs.py:53-56 freezes xyz / scale / rotation / opacity): nothing
    # densifies then and nobody reads the screen-space gradient (post_refine_gs.py:99,178-180 are commented out), and
    # without it the backward only has colour gradients to produce and takes its colour-only path.
    screenspace_points = _grad_sink(xyz, xyz.requires_grad)
    tanfovx = math.tan(viewpoint_camera.FoVx * 0.5)
    tanfovy = math.tan(viewpoint_camera.FoVy * 0.5)
    w, h = viewpoint_camera.resolution
    raster_settings = GaussianRasterizationSettings(
        image_height=int(h), image_width=int(w), tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=bool(getattr(pipe, "debug", False)))
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    # Fast path of the build's own harnesses: hand the model's raw leaves to the kernels (activations + SH concat fused
    # in). Same numbers as the branch below; taken only when no python-side colour / covariance branch is requested.
    if (override_color is None and _fused_activations(pipe, pc)
            and not getattr(pipe, "compute_cov3D_python", False) and not getattr(pipe, "convert_SHs_python", False)):
        f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
        rendered_image, radii, rendered_depth, rendered_alpha = rasterize_gaussians_raw(
            xyz, screenspace_points, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings)
        if getattr(pipe, "raw_outputs", False):   # scorp_amd.align scores poses from the un-normalised depth (one launch less per view)
            return {"render": rendered_image, "radii": radii, "render_depth_raw": rendered_depth, "render_alpha": rendered_alpha}
        rendered_depth, visible = render_tail(rendered_depth, rendered_alpha, radii)   # D / A with NaN -> 0, radii > 0
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": visible,
                "radii": radii, "render_depth": rendered_depth, "render_alpha": rendered_alpha}

    means3D, means2D, opacity = xyz, screenspace_points, pc.get_opacity
    scales = rotations = cov3D_precomp = None
    if getattr(pipe, "compute_cov3D_python", False):
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if getattr(pipe, "convert_SHs_python", False):
            shs_view = pc.get_features.transpose(1, 2).view(-1, 3, (pc.max_sh_degree + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(pc.get_features.shape[0], 1)
            dir_pp_normalized = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            sh2rgb = eval_sh(pc.active_sh_degree, shs_view, dir_pp_normalized)
            colors_precomp = torch.clamp_min(sh2rgb + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii, rendered_depth, rendered_alpha = rasterizer(
        means3D=means3D, means2D=means2D, shs=shs, colors_precomp=colors_precomp, opacities=opacity,
        scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    rendered_depth, visible = render_tail(rendered_depth, rendered_alpha, radii)
    return {
        "render": rendered_image,
        "viewspace_points": screenspace_points,
        "visibility_filter": visible,
        "radii": radii,
        "render_depth": rendered_depth,
        "render_alpha": rendered_alpha,
    }
