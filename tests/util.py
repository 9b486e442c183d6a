"""Shared helpers for the parity tests: seeded scenes in the rasterizer's argument layout."""
import math

import numpy as np

from scorp_amd.synthetic import activate, make_gaussians, ring_cameras


def make_case(N, W, H, deg, seed, log_scale=math.log(0.05), log_scale_std=0.6, radius=4.0, max_deg=3,
              precomp_color=False, precomp_cov=False, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, cam_index=None):
    """Returns (kw, cam): kw holds numpy float32 arrays + scalars accepted by both OracleRender and the HIP path."""
    raw = make_gaussians(N, max_deg, seed, log_scale_mean=log_scale, log_scale_std=log_scale_std)
    act = activate(raw)
    cams = ring_cameras(7, W, H, seed, radius=radius)
    cam = cams[(seed if cam_index is None else cam_index) % 7]
    kw = dict(means3D=act["means3D"], opacities=act["opacities"], W=W, H=H,
              tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
              view=cam.world_view_transform.numpy().astype(np.float32),
              proj=cam.full_proj_transform.numpy().astype(np.float32),
              campos=cam.camera_center.numpy().astype(np.float32), bg=np.asarray(bg, np.float32))
    rng = np.random.default_rng(seed + 77)
    if precomp_color:
        kw["colors_precomp"] = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    else:
        kw["shs"] = act["shs"]
        kw["sh_degree"] = deg
    if precomp_cov:
        Lm = rng.normal(0, math.exp(log_scale), (N, 3, 3))
        S = Lm @ Lm.transpose(0, 2, 1)
        kw["cov3D_precomp"] = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    else:
        kw["scales"] = act["scales"]
        kw["rotations"] = act["rotations"]
        kw["scale_modifier"] = scale_modifier
    return kw, cam


def image_weights(H, W, seed):
    """Seeded upstream gradients for (color, depth, alpha)."""
    rng = np.random.default_rng(seed + 4242)
    return (rng.normal(0, 1, (3, H, W)).astype(np.float32), rng.normal(0, 1, (1, H, W)).astype(np.float32),
            rng.normal(0, 1, (1, H, W)).astype(np.float32))
