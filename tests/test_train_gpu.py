"""The optimisation loops (train_3dgs / post_refine counterparts) on small synthetic scenes: quality must improve."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _teacher_student(dev, n, deg, seed):
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.synthetic import make_gaussians
    raw = make_gaussians(n, deg, seed, extent=1.0, log_scale_mean=math.log(0.05))
    raw["opacity"] += 1.5
    teacher = GaussianModel.from_raw(raw, deg, device=dev)
    teacher.active_sh_degree = deg
    rng = np.random.default_rng(seed + 1)
    raw2 = {k: v.copy() for k, v in raw.items()}
    raw2["features_dc"] += rng.normal(0, 0.6, raw2["features_dc"].shape).astype(np.float32)
    return teacher, raw2


def test_training_loop_improves_psnr_and_densifies(dev):
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.synthetic import ring_cameras
    from scorp_amd.train import PipelineParams, evaluate_psnr, render_views_gt, train
    teacher, raw2 = _teacher_student(dev, 4000, 1, 3)
    raw2["xyz"] += np.random.default_rng(9).normal(0, 0.01, raw2["xyz"].shape).astype(np.float32)
    cams = ring_cameras(8, 160, 120, 4, radius=3.0, device=dev)
    gts = render_views_gt(teacher, cams)
    student = GaussianModel.from_raw(raw2, 1, device=dev)
    student.active_sh_degree = 1
    opt = OptimizationParams()
    opt.densify_from_iter, opt.densification_interval, opt.opacity_reset_interval = 90, 100, 10_000
    opt.random_background = False
    p0 = evaluate_psnr(student, cams, gts)
    n0 = student.get_xyz.shape[0]
    losses = train(student, cams, gts, opt, PipelineParams(), iterations=320, scene_extent=3.0)
    p1 = evaluate_psnr(student, cams, gts)
    assert all(math.isfinite(v) for v in losses)
    assert p1 > p0 + 1.0, (p0, p1)
    assert student.get_xyz.shape[0] != n0            # densify_and_prune ran (iterations 100, 200, 300)


def test_post_refine_only_moves_colours(dev):
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.synthetic import ring_cameras
    from scorp_amd.train import evaluate_psnr, post_refine, render_views_gt
    teacher, raw2 = _teacher_student(dev, 3000, 0, 5)
    cams = ring_cameras(6, 128, 128, 6, radius=3.0, device=dev)
    gts, alphas = render_views_gt(teacher, cams, with_alpha=True)
    student = GaussianModel.from_raw(raw2, 0, device=dev)
    before = {n: getattr(student, n).detach().clone() for n in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc")}
    p0 = evaluate_psnr(student, cams, gts)
    losses = post_refine(student, cams, gts, [(a > 0.5).float() for a in alphas], OptimizationParams(), iterations=150)
    p1 = evaluate_psnr(student, cams, gts)
    assert p1 > p0 + 2.0, (p0, p1)
    assert losses[-1] < losses[0]
    for n in ("_xyz", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(student, n).detach(), before[n]), n
    assert not torch.equal(student._features_dc.detach(), before["_features_dc"])
