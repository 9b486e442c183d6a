"""Dev soak: a few thousand fused-view training iterations with densification on a mid-size synthetic scene (3DGS and
2DGS): finishes, finite losses, PSNR up, memory flat.  python scripts/dev/soak_train.py [iterations]"""
import math, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from scorp_amd.gaussian_model import GaussianModel, OptimizationParams, OptimizationParams2D
from scorp_amd.renderer2d import GaussianModel2D
from scorp_amd.synthetic import make_gaussians, ring_cameras
from scorp_amd.train import PipelineParams, train
from scorp_amd.renderer import render as render3d
from scorp_amd.renderer2d import render as render2d


def views(model, cams, rf, pipe):
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():
        return [rf(c, model, pipe, bg)["render"].clamp(0, 1).clone() for c in cams]


def psnr_of(model, cams, gts, rf, pipe):
    v = views(model, cams, rf, pipe)
    return float(torch.stack([-10.0 * torch.log10(((a - b) ** 2).mean()) for a, b in zip(v, gts)]).mean())
dev = torch.device('cuda:0')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2900   # not a multiple of the opacity reset interval: the last reset has 900 iterations to recover
for surfels in (False, True):
    Model = GaussianModel2D if surfels else GaussianModel
    N, deg = int(os.environ.get('SOAK_N', 150_000)), 2
    raw = make_gaussians(N, deg, 11, log_scale_mean=math.log(0.012))
    if surfels:
        raw["scaling"] = raw["scaling"][:, :2].copy()
    teacher = Model.from_raw(raw, deg, device=dev); teacher.active_sh_degree = deg
    raw2 = {k: v.copy() for k, v in raw.items()}
    rng = np.random.default_rng(5)
    raw2["xyz"] += rng.normal(0, 0.01, raw2["xyz"].shape).astype(np.float32)
    raw2["features_dc"] += rng.normal(0, 0.3, raw2["features_dc"].shape).astype(np.float32)
    student = Model.from_raw(raw2, deg, device=dev); student.active_sh_degree = deg
    W_, H_ = (int(v) for v in os.environ.get('SOAK_WH', '800x600').split('x'))
    cams = ring_cameras(24, W_, H_, 4, radius=3.5, device=dev)
    kw = dict(surfels=True) if surfels else {}
    rf = render2d if surfels else render3d
    pipe = PipelineParams(); pipe.fused_activations = True
    gts = views(teacher, cams, rf, pipe)
    opt = OptimizationParams2D() if surfels else OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval = 100, 100, 1000
    opt.random_background = False
    opt.opacity_cull, opt.max_screen_size = 0.005, 20
    p0 = psnr_of(student, cams, gts, rf, pipe)
    torch.cuda.synchronize(); m0 = torch.cuda.memory_allocated(); t0 = time.perf_counter()
    losses = train(student, cams, gts, opt, pipe, iterations=iters, scene_extent=3.0, fused_view=True, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    p1 = psnr_of(student, cams, gts, rf, pipe)
    print(("2DGS" if surfels else "3DGS"), "iters", iters, "it/s", round(iters / dt, 1), "N", N, "->", student.get_xyz.shape[0],
          "psnr", round(p0, 2), "->", round(p1, 2), "loss", round(float(losses[0]), 4), "->", round(float(losses[-1]), 4),
          "finite", all(math.isfinite(float(v)) for v in losses), "mem MB", round(m0 / 1e6), "->", round(torch.cuda.memory_allocated() / 1e6),
          "peak", round(torch.cuda.max_memory_allocated() / 1e6), flush=True)
