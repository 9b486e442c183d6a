"""Dev: the reference's training iteration, statement for statement (train_3dgs.py:94-193 without densification / logging), around
this package's imports - what a script gets with no edit beyond the imports of INTEGRATION.md section 1:
python scripts/dev/time_dropin_train.py [iterations]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.renderer import render
from scorp_amd.loss import l1_loss, ssim

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
gaussians = GaussianModel.from_raw(make_gaussians(1_000_000, 3, 11), 3, device=dev); gaussians.active_sh_degree = 3
opt = OptimizationParams()
gaussians.training_setup(opt)
cams = ring_cameras(8, 1600, 1200, 4, device=dev)
class pipe: convert_SHs_python = False; compute_cov3D_python = False; debug = False     # the reference's PipelineParams
background = torch.zeros(3, device=dev)
with torch.no_grad():
    gts = [render(c, gaussians, pipe, background)["render"].clamp(0, 1).clone() for c in cams]


def iteration(it):
    gaussians.update_learning_rate(it)
    viewpoint_cam = cams[it % 8]
    render_pkg = render(viewpoint_cam, gaussians, pipe, background)
    image, viewspace_point_tensor, visibility_filter, radii = (render_pkg["render"], render_pkg["viewspace_points"],
                                                                render_pkg["visibility_filter"], render_pkg["radii"])
    gt_image = gts[it % 8]
    Ll1 = l1_loss(image, gt_image)
    loss = (1.0 - opt.lambda_dssim) * Ll1 + opt.lambda_dssim * (1.0 - ssim(image, gt_image))
    loss.backward()
    with torch.no_grad():
        if not os.environ.get("NO_STATS"):
            gaussians.max_radii2D[visibility_filter] = torch.max(gaussians.max_radii2D[visibility_filter], radii[visibility_filter])
            gaussians.add_densification_stats(viewspace_point_tensor, visibility_filter)
        gaussians.optimizer.step()
        gaussians.optimizer.zero_grad(set_to_none=True)


for i in range(20):
    iteration(i + 1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(iters):
    iteration(21 + i)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("unedited training iteration: it/s", round(iters / dt, 1), "ms/it", round(1e3 * dt / iters, 3))
