"""Per-kernel event times of one training view (debug aid): `python scripts/time_pergaussian.py [S3|S6]`.
SCORP_GS_LIB selects a variant library (scripts/build_variant.sh) for same-box A/B runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scorp_amd import _C
from scorp_amd.fused_loss import fused_l1_ssim_loss
from scorp_amd.synthetic import SCENES, make_gaussians, ring_cameras

scene = sys.argv[1] if len(sys.argv) > 1 else "S3"
surfels = scene == "S6"
if surfels:
    from scorp_amd.renderer2d import GaussianModel2D as GaussianModel, render
else:
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.renderer import render


class Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = True
    depth_ratio = 0.0


dev = torch.device("cuda:0")
N, W, H, deg, seed, ncam = SCENES[scene]
model = GaussianModel.from_raw(make_gaussians(N, deg, seed, scale_dims=2 if surfels else 3), deg, device=dev)
model.active_sh_degree = deg
cams = ring_cameras(ncam, W, H, seed, device=dev)[:8]
bg = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render(c, model, Pipe(), bg)["render"].clone() for c in cams]
params = [model._xyz, model._features_dc, model._features_rest, model._scaling, model._rotation, model._opacity]


def step(i):
    out = render(cams[i % 8], model, Pipe(), bg)
    loss = fused_l1_ssim_loss(out["render"], gts[i % 8], 0.2)
    loss.backward()
    for p in params:
        p.grad = None


for i in range(4):
    step(i)
torch.cuda.synchronize()
_C.prof_enable(True)
for i in range(48):
    step(i)
torch.cuda.synchronize()
k = _C.prof_collect()
_C.prof_enable(False)
print(scene, os.path.basename(os.environ.get("SCORP_GS_LIB", "default")),
      {n: round(1e3 * ms / max(cnt, 1), 1) for n, (ms, cnt) in k.items() if ("preprocess" in n or "blend" in n) and cnt})
