"""TEST INFRASTRUCTURE — PyTorch restatement of the per-pixel tail of the 2DGS render().

Follows gs2dgs/gaussian_renderer/__init__.py:131-160 (alpha / world-space normal / nan_to_num'd expected and median
depth / depth_ratio mix / surf_normal * alpha.detach()) and gs2dgs/utils/point_utils.py:9-40 (depths_to_points,
depth_to_normal) op for op, on whatever device the inputs live on.  PARITY UNPINNED against a run of the reference:
its point_utils module hard-codes `.cuda()` and imports cv2 / matplotlib, none of which exist in the build container;
the code is plain tensor algebra and is restated line by line.  Only tests may import this.
"""
import torch


def camera_rays(world_view_transform, full_proj_transform, W, H):
    """rays_d[H*W,3], rays_o[3] (point_utils.py:10-22)."""
    dev = world_view_transform.device
    c2w = (world_view_transform.T).inverse()
    ndc2pix = torch.tensor([[W / 2, 0, 0, W / 2], [0, H / 2, 0, H / 2], [0, 0, 0, 1]]).float().to(dev).T
    projection_matrix = c2w.T @ full_proj_transform
    intrins = (projection_matrix @ ndc2pix)[:3, :3].T
    grid_x, grid_y = torch.meshgrid(torch.arange(W, device=dev).float(), torch.arange(H, device=dev).float(), indexing="xy")
    points = torch.stack([grid_x, grid_y, torch.ones_like(grid_x)], dim=-1).reshape(-1, 3)
    rays_d = points @ intrins.inverse().T @ c2w[:3, :3].T
    rays_o = c2w[:3, 3]
    return rays_d, rays_o


def depth_to_normal(rays_d, rays_o, depth):
    """point_utils.py:24-40 with the ray table passed in; depth [1,H,W] -> [H,W,3]."""
    points = (depth.reshape(-1, 1) * rays_d + rays_o).reshape(*depth.shape[1:], 3)
    output = torch.zeros_like(points)
    dx = points[2:, 1:-1] - points[:-2, 1:-1]
    dy = points[1:-1, 2:] - points[1:-1, :-2]
    normal_map = torch.nn.functional.normalize(torch.cross(dx, dy, dim=-1), dim=-1)
    output[1:-1, 1:-1, :] = normal_map
    return output


def surfel_maps_ref(allmap, world_view_transform, rays_d, rays_o, depth_ratio):
    """(render_alpha, render_normal, render_dist, surf_depth, surf_normal) as __init__.py:131-160 builds them."""
    render_alpha = allmap[1:2]
    render_normal = allmap[2:5]
    render_normal = (render_normal.permute(1, 2, 0) @ (world_view_transform[:3, :3].T)).permute(2, 0, 1)
    render_depth_median = torch.nan_to_num(allmap[5:6], 0, 0)
    render_depth_expected = torch.nan_to_num(allmap[0:1] / render_alpha, 0, 0)
    render_dist = allmap[6:7]
    surf_depth = render_depth_expected * (1 - depth_ratio) + depth_ratio * render_depth_median
    surf_normal = depth_to_normal(rays_d, rays_o, surf_depth).permute(2, 0, 1)
    surf_normal = surf_normal * render_alpha.detach()
    return render_alpha, render_normal, render_dist, surf_depth, surf_normal
