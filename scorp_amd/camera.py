"""Camera matrices for the splat renderer.

Mirrors the fields `render()` reads from the reference's `Camera` / `MiniCam`
(gs3dgs/scene/cameras.py:27-97,172-193): `FoVx`, `FoVy`, `resolution=(w,h)`, `world_view_transform`,
`full_proj_transform`, `camera_center`, `znear=.01`, `zfar=100`.  Matrix conventions follow
gs3dgs/utils/graphics_utils.py:38-71: matrices are stored transposed (row-vector convention), so a flat
view of `world_view_transform` is the column-major layout of the maths matrix — which is what the C ABI
expects (include/scorp_gs.h).
"""
import math

import numpy as np
import torch


def getWorld2View2(R, t, translate=np.array([0.0, 0.0, 0.0]), scale=1.0):
    """World->view 4x4 (maths convention) from a camera-to-world rotation R and a world->camera translation t
    (graphics_utils.py:38-49)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R).transpose()
    Rt[:3, 3] = t
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + translate) * scale
    return np.float32(np.linalg.inv(C2W))


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Perspective matrix, z in [0,1], w = +z (graphics_utils.py:51-71)."""
    tx, ty = math.tan(fovX / 2), math.tan(fovY / 2)
    top, right = ty * znear, tx * znear
    bottom, left = -top, -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov, pixels):
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal, pixels):
    return 2 * math.atan(pixels / (2 * focal))


class Camera:
    """The subset of the reference `Camera` the render path touches (cameras.py:62-97,139-170)."""

    def __init__(self, R, T, FoVx, FoVy, resolution, device="cpu", uid=0, image_name=None,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0):
        self.uid = uid
        self.R, self.T = np.asarray(R, np.float64), np.asarray(T, np.float64)
        self.FoVx, self.FoVy = float(FoVx), float(FoVy)
        self.FoVx_old, self.FoVy_old = self.FoVx, self.FoVy     # cameras.py:70-71: what restore_fov() goes back to
        self.image_name = image_name
        self.resolution = (int(resolution[0]), int(resolution[1]))
        self.resolution_original = self.resolution
        self.image_width, self.image_height = self.resolution
        self.zfar, self.znear = 100.0, 0.01
        self.trans, self.scale = trans, scale
        self.device = torch.device(device)
        self._update()

    def _update(self):
        self.world_view_transform = torch.tensor(getWorld2View2(self.R, self.T, self.trans, self.scale)).transpose(0, 1).contiguous().to(self.device)   # contiguous: the rasterizer takes it without a per-view copy kernel
        self.projection_matrix = getProjectionMatrix(self.znear, self.zfar, self.FoVx, self.FoVy).transpose(0, 1).contiguous().to(self.device)
        self.full_proj_transform = self.world_view_transform.unsqueeze(0).bmm(self.projection_matrix.unsqueeze(0)).squeeze(0)
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()

    def to(self, device):
        self.device = torch.device(device)
        self._update()
        return self

    def scale_resolution(self, scale=1.0):
        """cameras.py:139-148 — the align loop renders at up to 1.5^3 x the base resolution."""
        self.resolution = (int(self.resolution[0] * scale), int(self.resolution[1] * scale))
        self.image_width, self.image_height = self.resolution

    def restore_resolution(self):
        """cameras.py:147-148."""
        self.resolution = self.resolution_original
        self.image_width, self.image_height = self.resolution

    def scale_fov(self, scale_x, scale_y):
        """cameras.py:150-151: both fields of view scaled, projection matrices rebuilt."""
        self._update_fov(self.FoVx * scale_x, self.FoVy * scale_y)

    def restore_fov(self):
        """cameras.py:169-170."""
        self._update_fov(self.FoVx_old, self.FoVy_old)

    def _update_fov(self, fovx, fovy):
        """cameras.py:153-167 (the reference rebuilds projection_matrix and full_proj_transform; so does _update)."""
        self.FoVx, self.FoVy = float(fovx), float(fovy)
        self._update()


def look_at_camera(position, target, up, FoVx, resolution, device="cpu", uid=0):
    """COLMAP-style camera (+x right, +y down, +z forward) at `position` looking at `target`."""
    position, target, up = (np.asarray(v, np.float64) for v in (position, target, up))
    fwd = target - position
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right)
    down = np.cross(fwd, right)
    W2C = np.stack([right, down, fwd])          # rows
    R = W2C.T                                   # camera-to-world, as the reference stores it
    T = -W2C @ position
    w, h = resolution
    FoVy = focal2fov(fov2focal(FoVx, w), h)
    return Camera(R, T, FoVx, FoVy, resolution, device=device, uid=uid)
