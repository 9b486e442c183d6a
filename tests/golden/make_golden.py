"""Generate tests/golden/ref_helpers.npz by IMPORTING the reference's CPU-importable helpers.

Run in the build container only (the GPU box has no /root/reference):
    PYTHONPATH=/root/reference python tests/golden/make_golden.py
Only inputs and outputs are stored (SURVEY.md §8c, G1–G8); no reference source text is copied.
The renderer itself cannot be captured: `diff_gaussian_rasterization` is absent from the reference tree.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gs3dgs.utils.sh_utils import eval_sh, RGB2SH, SH2RGB  # noqa: E402
from gs3dgs.utils.graphics_utils import getProjectionMatrix, getWorld2View2, fov2focal, focal2fov  # noqa: E402
from gs3dgs.utils.loss_utils import l1_loss, ssim, isotropic_loss  # noqa: E402
from gs3dgs.utils.image_utils import psnr, depth_normalize_  # noqa: E402
from gs3dgs.utils.general_utils import get_expon_lr_func, inverse_sigmoid  # noqa: E402
from utils.geometry import quaternion_to_matrix_tensor  # noqa: E402
from utils.solution import kabsch_algorithm_np, umeyama_algorithm_np  # noqa: E402

out = {}
rng = np.random.default_rng(20260630)

# G1: eval_sh, degrees 0..3, layout sh[N,3,K] as the python branch of render() uses it
sh = rng.normal(0, 0.5, (64, 3, 16)).astype(np.float32)
dirs = rng.normal(0, 1, (64, 3)).astype(np.float32)
dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
out["g1_sh"], out["g1_dirs"] = sh, dirs
for deg in range(4):
    rgb = eval_sh(deg, torch.tensor(sh), torch.tensor(dirs))
    out[f"g1_rgb_deg{deg}"] = rgb.numpy()
    out[f"g1_clamped_deg{deg}"] = torch.clamp_min(rgb + 0.5, 0.0).numpy()
out["g1_rgb2sh"] = RGB2SH(torch.tensor(sh[:, :, 0])).numpy()
out["g1_sh2rgb"] = SH2RGB(torch.tensor(sh[:, :, 0])).numpy()

# G2: projection matrices and world->view matrices (R from the committed rotation fixture)
rots = np.load("/root/reference/rotation_matrices/rotations_32.npz")["rotations"]
fovs = np.array([[1.0471976, 0.8], [0.6, 0.6], [1.4, 1.1], [0.3, 0.5]])
out["g2_fovs"] = fovs
out["g2_proj"] = np.stack([getProjectionMatrix(0.01, 100.0, fx, fy).numpy() for fx, fy in fovs])
ts = rng.normal(0, 2, (4, 3))
out["g2_R"], out["g2_t"] = rots[:4], ts
out["g2_w2v"] = np.stack([getWorld2View2(rots[i], ts[i]) for i in range(4)])
out["g2_w2v_ts"] = np.stack([getWorld2View2(rots[i], ts[i], np.array([0.1, -0.2, 0.3]), 1.5) for i in range(4)])
out["g2_focal"] = np.array([fov2focal(1.0471976, 1600), focal2fov(1385.64, 1200)])

# G3: L1 / SSIM / PSNR with gradients
a = rng.uniform(0, 1, (3, 64, 80)).astype(np.float32)
b = np.clip(a + rng.normal(0, 0.1, a.shape), 0, 1).astype(np.float32)
ta = torch.tensor(a, requires_grad=True)
tb = torch.tensor(b)
l1 = l1_loss(ta, tb)
s = ssim(ta, tb)
loss = 0.8 * l1 + 0.2 * (1.0 - s)
loss.backward()
out["g3_a"], out["g3_b"] = a, b
out["g3_l1"], out["g3_ssim"], out["g3_loss"] = l1.item(), s.item(), loss.item()
out["g3_grad_a"] = ta.grad.numpy()
out["g3_psnr"] = psnr(torch.tensor(a), torch.tensor(b)).numpy()

# G4: exponential LR schedule of the xyz group
f = get_expon_lr_func(1.6e-4, 1.6e-6, 0, 0.01, 30000)
steps = np.array([0, 1, 100, 15000, 30000])
out["g4_steps"], out["g4_lr"] = steps, np.array([f(int(s_)) for s_ in steps])
out["g4_inv_sigmoid"] = inverse_sigmoid(torch.tensor([0.1, 0.5, 0.9])).numpy()

# G5/G7: quaternion (w,x,y,z) -> rotation; same formula as general_utils.build_rotation (cuda-only there)
q = rng.normal(0, 1, (32, 4)).astype(np.float32)
out["g5_q"] = q
out["g5_R"] = quaternion_to_matrix_tensor(torch.tensor(q)).numpy()

# G6: Kabsch / Umeyama fits of seeded 50-point sets (a planted similarity + noise; one pair with a reflection-prone SVD)
src = rng.normal(0, 1, (4, 50, 3))
tgt = np.empty_like(src)
for i in range(4):
    A = rng.normal(0, 1, (3, 3))
    U, _, Vt = np.linalg.svd(A)
    R0 = U @ Vt
    if np.linalg.det(R0) < 0:
        R0[:, 0] *= -1
    tgt[i] = (0.5 + i) * (src[i] @ R0.T) + rng.normal(0, 1, 3) + rng.normal(0, 0.02 * (1 + 20 * (i == 3)), (50, 3))
out["g6_src"], out["g6_tgt"] = src, tgt
ks = [kabsch_algorithm_np(src[i], tgt[i]) for i in range(4)]
us = [umeyama_algorithm_np(src[i], tgt[i]) for i in range(4)]
out["g6_kabsch_R"], out["g6_kabsch_t"] = np.stack([k[0] for k in ks]), np.stack([k[1] for k in ks])
out["g6_umeyama_R"], out["g6_umeyama_t"] = np.stack([u[0] for u in us]), np.stack([u[1] for u in us])
out["g6_umeyama_s"] = np.array([u[2] for u in us])

# G8: the regularisation helpers of the late training iterations
sc = np.exp(rng.normal(-4, 0.5, (200, 3))).astype(np.float32)
out["g8_scaling"], out["g8_isotropic"] = sc, isotropic_loss(torch.tensor(sc)).item()
dm = rng.uniform(0.5, 6.0, (300,)).astype(np.float32)
out["g8_depth"], out["g8_depth_normalized"] = dm, depth_normalize_(torch.tensor(dm)).numpy()

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_helpers.npz")
np.savez_compressed(dst, **out)
print("wrote", dst, {k: np.asarray(v).shape for k, v in out.items()})
