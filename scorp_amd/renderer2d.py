"""`render()` of the 2DGS path — host-side mirror of gs2dgs/gaussian_renderer/__init__.py:24-170 — and the surfel
flavour of `GaussianModel` (gs2dgs/scene/gaussian_model.py: 2-D scales :49,136, random initial rotations :137,
4x4 splat2world `get_covariance` :27-33, densification statistic over all three components :495).

Result dict keys as the reference returns them: render, viewspace_points, visibility_filter, radii, render_alpha,
render_normal (rotated to world space), render_dist, render_depth (= surf_depth), surf_normal.
"""
import math

import numpy as np
import torch
import torch.nn as nn

from .gaussian_model import GaussianModel, build_rotation, inverse_sigmoid
from .rasterizer2d import GaussianRasterizer, rasterize_surfels_raw, surfel_maps, surfel_regularizer_losses
from .rasterizer3d import GaussianRasterizationSettings
from .renderer import _fused_activations
from .sh import RGB2SH, eval_sh



def _camera_rays(view, dev):
    """Per-pixel ray directions / origin of gs2dgs/utils/point_utils.py:9-22.  They depend only on the camera and its
    current resolution, so they are built once per (camera, resolution) and cached on the camera object — the
    reference rebuilds them (two 4x4 inversions, a host->device copy, a meshgrid) on every render."""
    W, H = view.resolution
    wv, fp = view.world_view_transform, view.full_proj_transform
    # identity AND version of the two matrices: an in-place pose update (same storage) bumps `_version`, a replaced
    # tensor changes the id even if the allocator hands back the old address
    key = (W, H, str(dev), id(wv), wv._version, id(fp), fp._version)
    cache = getattr(view, "_scorp_rays", None)
    if cache is not None and cache[0] == key:
        return cache[1], cache[2]
    c2w = (view.world_view_transform.T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]], device=dev).float().T
    projection_matrix = c2w.T @ view.full_proj_transform
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(), indexing="xy")
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = (points @ intrins.inverse().T @ c2w[:3, :3].T).contiguous()
    rays_o = c2w[:3, 3].contiguous()
    try:
        view._scorp_rays = (key, rays_d, rays_o, wv, fp)   # (holding the tensors keeps their ids from being reused)
    except Exception:
        pass
    return rays_d, rays_o


def depths_to_points(view, depthmap):
    """gs2dgs/utils/point_utils.py:9-24, on the depth map's device."""
    rays_d, rays_o = _camera_rays(view, depthmap.device)
    return depthmap.reshape(-1, 1) * rays_d + rays_o


def depth_to_normal(view, depth):
    """Pseudo surface normal from a depth map (point_utils.py:26-37)."""
    points = depths_to_points(view, depth).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    output[1:-1, 1:-1, :] = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    return output


_ZEROS = {}


def _grad_sink(xyz):
    """A fresh leaf of zeros shaped like xyz (the screen-space gradient sink) over a cached, never-written storage."""
    key = (xyz.shape[0], xyz.dtype, xyz.device)
    z = _ZEROS.get(key)
    if z is None:
        if len(_ZEROS) > 8:
            _ZEROS.clear()
        z = _ZEROS[key] = torch.zeros_like(xyz, requires_grad=False)
    return z.detach().requires_grad_(True)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
    xyz = pc.get_xyz
    # gradient sink for the screen-space means; the reference builds it as `zeros_like(...) + 0` with retain_grad()
    # (gaussian_renderer/__init__.py:39-43) — a leaf with requires_grad gives the caller the same `.grad` without the
    # extra 12 MB add kernel per view; the zeros themselves are shared between views (nothing ever writes them), so
    # not even a fill kernel runs: every view gets a fresh leaf over the same storage
    screenspace_points = _grad_sink(xyz)
    tanfovx, tanfovy = math.tan(viewpoint_camera.FoVx * 0.5), math.tan(viewpoint_camera.FoVy * 0.5)
    w, h = viewpoint_camera.resolution
    raster_settings = GaussianRasterizationSettings(
        image_height=int(h), image_width=int(w), tanfovx=tanfovx, tanfovy=tanfovy, bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)

    fused = (override_color is None and _fused_activations(pipe, pc)
             and not getattr(pipe, "compute_cov3D_python", False))
    if fused:
        f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw = pc.raw_leaves()
        rendered_image, radii, allmap = rasterize_surfels_raw(xyz, screenspace_points, f_dc, f_rest, opacity_raw,
                                                              scaling_raw, rotation_raw, raster_settings)
    else:
        scales = rotations = cov3D_precomp = None
        if getattr(pipe, "compute_cov3D_python", False):
            splat2world = pc.get_covariance(scaling_modifier)
            W, H = viewpoint_camera.resolution
            near, far = viewpoint_camera.znear, viewpoint_camera.zfar
            ndc2pix = torch.tensor([[W / 2, 0, 0, (W - 1) / 2], [0, H / 2, 0, (H - 1) / 2], [0, 0, far - near, near],
                                    [0, 0, 0, 1]], device=xyz.device).float().T
            world2pix = viewpoint_camera.full_proj_transform @ ndc2pix
            cov3D_precomp = (splat2world[:, [0, 1, 3]] @ world2pix[:, [0, 1, 3]]).permute(0, 2, 1).reshape(-1, 9)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
        shs = colors_precomp = None
        if override_color is None:
            shs = pc.get_features            # the reference forces convert_SHs_python = False (:96)
        else:
            colors_precomp = override_color
        rendered_image, radii, allmap = GaussianRasterizer(raster_settings=raster_settings)(
            means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp, opacities=pc.get_opacity,
            scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)

    rets = RenderPackage({"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                          "radii": radii})
    # the per-pixel tail (gs2dgs/gaussian_renderer/__init__.py:131-160: world-space normals, nan_to_num'd expected /
    # median depth mixed by depth_ratio, pseudo surface normal of that depth times alpha.detach()) is one HIP kernel
    rays_d, rays_o = _camera_rays(viewpoint_camera, allmap.device)
    render_alpha, render_normal, render_dist, surf_depth, surf_normal = surfel_maps(
        allmap, viewpoint_camera.world_view_transform, rays_d, rays_o, getattr(pipe, "depth_ratio", 1.0))
    rets.update({"render_alpha": render_alpha, "render_normal": render_normal, "render_dist": render_dist,
                 "render_depth": surf_depth, "surf_normal": surf_normal})
    # what the fused regulariser needs, kept off the reference's nine keys
    rets.allmap, rets.camera, rets.depth_ratio = allmap, viewpoint_camera, getattr(pipe, "depth_ratio", 1.0)
    return rets


class GaussianModel2D(GaussianModel):
    """Surfel parameters: _scaling is [N,2]; everything else as the 3DGS container."""

    @classmethod
    def from_raw(cls, raw, sh_degree, device="cuda"):
        assert raw["scaling"].shape[1] == 2, "surfels have two scales"
        return super().from_raw(raw, sh_degree, device)

    def get_covariance(self, scaling_modifier=1):
        """4x4 splat2world, transposed storage (gs2dgs/scene/gaussian_model.py:27-33)."""
        s = torch.cat([self.get_scaling * scaling_modifier, torch.ones_like(self.get_scaling[:, :1])], dim=-1)
        RS = (build_rotation(self._rotation) * s[:, None, :]).permute(0, 2, 1)
        trans = torch.zeros((self._xyz.shape[0], 4, 4), dtype=torch.float, device=self._xyz.device)
        trans[:, :3, :3] = RS
        trans[:, 3, :3] = self._xyz
        trans[:, 3, 3] = 1
        return trans

    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        from simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(self.device)
        col = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(self.device))
        K = (self.max_sh_degree + 1) ** 2
        feats = torch.zeros((col.shape[0], 3, K), device=self.device)
        feats[:, :3, 0] = col
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 2)
        rots = torch.rand((pts.shape[0], 4), device=self.device)
        opac = inverse_sigmoid(0.1 * torch.ones((pts.shape[0], 1), device=self.device))
        P = lambda t: nn.Parameter(t.contiguous().requires_grad_(True))
        self._xyz, self._scaling, self._rotation, self._opacity = P(pts), P(scales), P(rots), P(opac)
        self._features_dc = P(feats[:, :, 0:1].transpose(1, 2))
        self._features_rest = P(feats[:, :, 1:].transpose(1, 2))
        self.max_radii2D = torch.zeros(pts.shape[0], device=self.device)

    _stats_norm_components = 3     # the norm below runs over the WHOLE row (gs2dgs/scene/gaussian_model.py:494-495)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self._masked_stats_add(viewspace_point_tensor.grad, update_filter)   # (the whole row; see GaussianModel._masked_stats_add)

    # ---- the "tuning mask" of gs2dgs/scene/gaussian_model.py:60,210-222,498-508: the surfels present when
    # prepare_gs_tuning_mask() was called (the first `num_mask` rows: later ones are appended by densification) are held
    # fixed - their gradients are zeroed before the optimizer step and reset_opacity() leaves their opacity alone.  No
    # script of the reference calls these; they are here so that a caller that does finds them.
    _num_mask = 0

    def prepare_gs_tuning_mask(self):
        self._num_mask = len(self._xyz)

    def gs_tuning_mask_grad(self):
        if self._xyz.grad is not None:
            for p in (self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity):
                if p.grad is not None:
                    p.grad[: self._num_mask] *= 0.0

    def reset_opacity(self):
        op = self.get_opacity
        capped = torch.min(op[self._num_mask:], torch.ones_like(op[self._num_mask:]) * 0.01)
        self.replace_tensor_to_optimizer(inverse_sigmoid(torch.cat([op[: self._num_mask], capped])), "opacity")


class RenderPackage(dict):
    """The dict render() returns (the reference's nine keys) that also remembers the rasterizer's allmap and the camera,
    so that `fused_surfel_regularizers` can work from allmap directly."""
    allmap = None
    camera = None
    depth_ratio = 1.0


def fused_surfel_regularizers(render_pkg, lambda_normal, lambda_dist):
    """(normal_loss, dist_loss) of train_2dgs.py:142-150 as ONE kernel each way, straight from the rasterizer's allmap:
    the gradient maps of these two terms are scaled copies of render_normal / surf_normal / a constant, so nothing but
    allmap is read and nothing but its gradient is written.  Same values as `surfel_regularizers`."""
    if getattr(render_pkg, "allmap", None) is None:
        raise RuntimeError("fused_surfel_regularizers needs the package returned by scorp_amd.renderer2d.render")
    cam = render_pkg.camera
    rays_d, rays_o = _camera_rays(cam, render_pkg.allmap.device)
    out = surfel_regularizer_losses(render_pkg.allmap, cam.world_view_transform, rays_d, rays_o, render_pkg.depth_ratio,
                                    lambda_normal, lambda_dist)
    return out[0], out[1]


def surfel_regularizers(render_pkg, lambda_normal, lambda_dist):
    """Normal-consistency and depth-distortion terms of train_2dgs.py:142-150."""
    rend_normal, surf_normal, rend_dist = render_pkg["render_normal"], render_pkg["surf_normal"], render_pkg["render_dist"]
    normal_loss = lambda_normal * (1 - (rend_normal * surf_normal).sum(dim=0))[None].mean()
    dist_loss = lambda_dist * rend_dist.mean()
    return normal_loss, dist_loss
