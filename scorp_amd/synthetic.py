"""Seeded synthetic splat scenes and camera rings (SURVEY.md §8d, S1–S6).

All draws use numpy.random.default_rng(seed) in a fixed order so every rank / test sees the same scene.
Distributions: xyz ~ U([-1.5,1.5]^3); log-scale ~ N(log 0.004, 0.5^2) per axis; quaternion ~ N(0,I) normalised;
opacity logit ~ N(0,1.5^2); SH dc ~ U(-1,1), rest ~ N(0,0.1^2); cameras on a ring of radius 4 at height
U(-0.5,1.5), looking at the origin, FoVx = 60 deg.
"""
import math

import numpy as np

from .camera import look_at_camera

SCENES = {
    # name: (N, width, height, sh_degree, seed, n_cameras)
    "S1": (10_000, 256, 256, 0, 1, 1),
    "S2": (500_000, 1600, 1200, 3, 2, 280),
    "S3": (1_000_000, 1600, 1200, 3, 3, 280),
    "S4": (100_000, 800, 800, 0, 4, 15),
    "S6": (1_000_000, 1600, 1200, 3, 6, 280),
}


def make_gaussians(N, sh_degree, seed, extent=1.5, log_scale_mean=math.log(0.004), log_scale_std=0.5, scale_dims=3):
    """Raw (pre-activation) parameters in the reference's GaussianModel layout
    (gs3dgs/scene/gaussian_model.py:28-62): _xyz[N,3], _features_dc[N,1,3], _features_rest[N,K-1,3],
    _scaling[N,3] (log), _rotation[N,4] (w,x,y,z, un-normalised), _opacity[N,1] (logit)."""
    rng = np.random.default_rng(seed)
    K = (sh_degree + 1) ** 2
    xyz = rng.uniform(-extent, extent, (N, 3)).astype(np.float32)
    scaling = rng.normal(log_scale_mean, log_scale_std, (N, scale_dims)).astype(np.float32)
    rotation = rng.normal(0, 1, (N, 4)).astype(np.float32)
    opacity = rng.normal(0, 1.5, (N, 1)).astype(np.float32)
    f_dc = rng.uniform(-1, 1, (N, 1, 3)).astype(np.float32)
    f_rest = rng.normal(0, 0.1, (N, K - 1, 3)).astype(np.float32)
    return dict(xyz=xyz, scaling=scaling, rotation=rotation, opacity=opacity, features_dc=f_dc, features_rest=f_rest)


def activate(raw):
    """The GaussianModel property activations (gaussian_model.py:126-146) in numpy."""
    rot = raw["rotation"] / np.linalg.norm(raw["rotation"], axis=1, keepdims=True)
    return dict(
        means3D=raw["xyz"],
        scales=np.exp(raw["scaling"]),
        rotations=rot.astype(np.float32),
        opacities=(1.0 / (1.0 + np.exp(-raw["opacity"]))).astype(np.float32),
        shs=np.concatenate([raw["features_dc"], raw["features_rest"]], axis=1),
    )


def ring_cameras(n, width, height, seed, radius=4.0, fovx_deg=60.0, device="cpu"):
    rng = np.random.default_rng(seed + 1000)
    heights = rng.uniform(-0.5, 1.5, n)
    cams = []
    for i in range(n):
        th = 2 * math.pi * i / n
        pos = (radius * math.cos(th), radius * math.sin(th), heights[i])
        cams.append(look_at_camera(pos, (0, 0, 0), (0, 0, 1), math.radians(fovx_deg), (width, height), device=device, uid=i))
    return cams


def scene(name, device="cpu", n_cameras=None):
    N, W, H, deg, seed, ncam = SCENES[name]
    raw = make_gaussians(N, deg, seed, scale_dims=2 if name == "S6" else 3)
    cams = ring_cameras(n_cameras or ncam, W, H, seed, device=device)
    return raw, cams, deg
