"""GPU parity of the small helpers: distCUDA2 vs an exact k-d tree, FusedAdam vs torch.optim.Adam."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 1000, 20011])
def test_distCUDA2_matches_kdtree(n, dev):
    from scipy.spatial import cKDTree
    from simple_knn._C import distCUDA2
    rng = np.random.default_rng(n)
    pts = rng.normal(0, 1, (n, 3)).astype(np.float32)
    if n >= 1000:
        pts[10] = pts[11]                                  # duplicates: distance 0 counts
    got = distCUDA2(torch.tensor(pts, device=dev)).cpu().numpy()
    k = min(4, n)
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=k)
    d = np.asarray(d).reshape(n, k)[:, 1:]                  # drop self
    ref = (d ** 2).sum(1) / 3.0 if k > 1 else np.zeros(n)
    np.testing.assert_allclose(got, ref, rtol=2e-5, atol=1e-9)


def test_fused_adam_matches_torch_adam(dev):
    from scorp_amd.fused_adam import FusedAdam
    g = torch.Generator(device=dev).manual_seed(3)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (1000, 3), (1000, 4), (7,)]
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 0.005, 0.001, 0.01]
    mk = lambda: [torch.nn.Parameter(torch.randn(s, device=dev, generator=torch.Generator(device=dev).manual_seed(i)))
                  for i, s in enumerate(shapes)]
    pa, pb = mk(), mk()
    oa = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(pa, lrs)], lr=0.0, eps=1e-15)
    ob = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(pb, lrs)], lr=0.0, eps=1e-15)
    for it in range(5):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, device=dev, generator=g) * (10.0 ** (it - 2))
            gr[::3] = 0.0                                   # zero gradients (invisible splats) exercise eps = 1e-15
            a.grad, b.grad = gr.clone(), gr.clone()
        oa.step(); ob.step()
        if it == 2:
            ob.param_groups[0]["lr"] = oa.param_groups[0]["lr"] = 3e-5      # update_learning_rate mid-run
    for a, b in zip(pa, pb):
        assert (a - b).abs().max() <= 2e-6 * a.abs().max()
        sa, sb = oa.state[a], ob.state[b]
        assert int(sa["step"]) == int(sb["step"]) == 5
        ea, eb = sa["exp_avg"].cpu().numpy(), sb["exp_avg"].cpu().numpy()
        np.testing.assert_allclose(eb, ea, rtol=2e-6, atol=2e-6 * np.abs(ea).max())      # 1 ulp of the larger operand
        va, vb = sa["exp_avg_sq"].cpu().numpy(), sb["exp_avg_sq"].cpu().numpy()
        np.testing.assert_allclose(vb, va, rtol=2e-6, atol=2e-6 * np.abs(va).max())
    sd = ob.state_dict()                                    # torch-compatible state layout
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}


def test_gaussian_model_trains_with_fused_adam_and_densifies(dev):
    """create_from_pcd (distCUDA2) -> training_setup (FusedAdam) -> a few render/backward/step iterations with
    densification stats, densify_and_prune and reset_opacity: the reference loop's host logic on the GPU path."""
    import math
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.renderer import render
    from scorp_amd.synthetic import ring_cameras

    class Pcd:
        pass

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
        fused_activations = True

    rng = np.random.default_rng(0)
    pcd = Pcd()
    pcd.points = rng.uniform(-1, 1, (3000, 3)).astype(np.float32)
    pcd.colors = rng.uniform(0, 1, (3000, 3)).astype(np.float32)
    m = GaussianModel(3, device=dev)
    m.create_from_pcd(pcd, 1.0)
    assert torch.isfinite(m._scaling).all() and m._features_rest.shape == (3000, 15, 3)
    opt = OptimizationParams()
    m.training_setup(opt)
    cams = ring_cameras(4, 160, 120, 1, device=dev)
    bg = torch.zeros(3, device=dev)
    with torch.no_grad():                                   # ground truth = the initial model; then perturb the colours
        gts = [render(c, m, Pipe(), bg)["render"].clone() for c in cams]
        m._features_dc.add_(0.5 * torch.randn(m._features_dc.shape, device=dev, generator=torch.Generator(device=dev).manual_seed(1)))
    losses = []
    for it in range(1, 13):
        m.update_learning_rate(it)
        if it % 4 == 0:
            m.oneupSHdegree()
        cam, gt = cams[it % 4], gts[it % 4]
        out = render(cam, m, Pipe(), bg)
        loss = fused_l1_ssim_loss(out["render"], gt, opt.lambda_dssim)
        loss.backward()
        losses.append(loss.item())
        with torch.no_grad():
            vis = out["visibility_filter"]
            m.max_radii2D[vis] = torch.max(m.max_radii2D[vis], out["radii"][vis].float())
            m.add_densification_stats(out["viewspace_points"], vis)
            if it == 8:
                n0 = m.get_xyz.shape[0]
                m.densify_and_prune(1e-7, 0.005, 2.0, 20)
                assert m.get_xyz.shape[0] != n0
            if it == 10:
                m.reset_opacity()
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
    assert all(math.isfinite(v) for v in losses)
    assert min(losses[4:7]) < losses[0]          # optimisation makes progress before the forced densify / opacity reset


def _object_model(dev, n, deg, seed):
    import math
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.synthetic import make_gaussians
    raw = make_gaussians(n, deg, seed, extent=0.8, log_scale_mean=math.log(0.03))
    raw["xyz"][:, 0] *= 1.6          # an elongated, asymmetric cloud: rotations are distinguishable
    raw["xyz"][:, 2] *= 0.5
    raw["xyz"][: n // 3, 1] += 0.6
    m = GaussianModel.from_raw(raw, deg, device=dev)
    m.active_sh_degree = deg
    return m


def test_rotating_object_and_camera_together_is_invisible(dev):
    """gaussians_rotate (means, quaternions, SH bands 1-3) against the renderer: rotating the world and the camera by
    the same R must reproduce the image (view-dependent colour included)."""
    import os
    from scorp_amd.camera import Camera
    from scorp_amd.renderer import render
    from scorp_amd.synthetic import ring_cameras
    from scorp_amd.transforms import gaussians_rotate

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
        fused_activations = True

    rots = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotations_128.npz"))["rotations"]
    R = torch.tensor(rots[41], dtype=torch.float32)
    m = _object_model(dev, 4000, 3, 3)
    cam = ring_cameras(5, 200, 160, 2, radius=3.0, device=dev)[1]
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    with torch.no_grad():
        ref = render(cam, m, Pipe(), bg)
        gaussians_rotate(m, R.to(dev))
        cam2 = Camera(R.numpy().astype(np.float64) @ cam.R, cam.T, cam.FoVx, cam.FoVy, cam.resolution, device=dev)
        out = render(cam2, m, Pipe(), bg)
    assert (out["render"] - ref["render"]).abs().mean() < 2e-5
    assert (out["render_alpha"] - ref["render_alpha"]).abs().mean() < 2e-5
    assert (out["radii"] > 0).sum() == (ref["radii"] > 0).sum()


def test_rotation_sweep_recovers_the_planted_rotation(dev):
    """Config #3 in miniature: 128 hypotheses (the reference's rotations_128.npz), 6 cameras, forward-only renders."""
    import os
    from scorp_amd.align import render_views, rotation_sweep
    from scorp_amd.synthetic import ring_cameras
    from scorp_amd.transforms import gaussians_rotate
    import copy
    rots = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotations_128.npz"))["rotations"]
    m = _object_model(dev, 3000, 0, 5)
    cams = ring_cameras(6, 128, 128, 9, radius=3.0, device=dev)
    bg = torch.zeros(3, device=dev)
    planted = 93
    tgt_model = copy.copy(m)
    tgt_model._xyz, tgt_model._rotation = m._xyz.detach().clone(), m._rotation.detach().clone()
    tgt_model._features_rest = m._features_rest.detach().clone()
    gaussians_rotate(tgt_model, torch.tensor(rots[planted], dtype=torch.float32, device=dev), fix_center=True)
    targets = render_views(tgt_model, cams, bg)
    from scorp_amd.align import SweepPlan
    ids, fit, best = rotation_sweep(m, rots, cams, targets, bg)          # SH-0 object: cameras moved, views stacked
    assert ids.numel() == 128 and best == planted
    assert float(fit[planted, 0]) > -1e-5 and float(fit[:, 0].sort().values[-2]) < float(fit[planted, 0]) - 1e-4
    ids_e, fit_e, best_e = rotation_sweep(m, rots, cams, targets, bg, use_graph=False)   # object rotated, eager launches
    assert best_e == planted and torch.equal(ids, ids_e)
    # rotating the cameras instead of 3 000 positions and quaternions: float rounding, plus the odd pixel whose alpha crosses
    # 1/255 (its normalised depth then jumps from 0 to ~3: 3 / (6 x 128 x 128) = 3e-5 per such pixel)
    assert (fit - fit_e).abs().max() < 2e-4
    plan = SweepPlan(m, cams, targets, bg, use_graph=True, stacked=False)                # object rotated, HIP-graph replay
    assert plan.graph is not None
    ids_g, fit_g, best_g = rotation_sweep(m, rots, cams, targets, bg, plan=plan)
    assert best_g == planted and (fit_g - fit_e).abs().max() < 1e-6
    # the one-launch score (scorp_gs3d_pose_score_accumulate on the raw depth / alpha) against the torch formulation
    # on render()'s normalised outputs
    from scorp_amd.align import hypothesis_fitness
    from scorp_amd.renderer import render
    for j in (0, planted, 127):
        Rj = torch.tensor(rots[j], dtype=torch.float32, device=dev)
        fused = float(hypothesis_fitness(m, Rj, cams, targets, bg))
        plain = float(hypothesis_fitness(m, Rj, cams, targets, bg, render_fn=lambda *a: render(*a)))
        assert abs(fused - plain) <= 1e-5 * max(1.0, abs(plain)), (j, fused, plain)
    # the model itself is untouched by the sweep
    assert torch.equal(tgt_model._features_dc, m._features_dc)


def test_render_tail_matches_torch(dev):
    """scorp_gs3d_render_tail (render_depth = nan_to_num(depth / alpha, 0, 0), visibility = radii > 0) and its backward
    against the torch ops of gs3dgs/gaussian_renderer/__init__.py:113-120, including empty pixels (0 / 0)."""
    from scorp_amd.rasterizer3d import render_tail
    g = torch.Generator(device=dev).manual_seed(3)
    H, W, N = 37, 61, 1000
    alpha = torch.rand(1, H, W, device=dev, generator=g)
    depth = (2.0 + torch.rand(1, H, W, device=dev, generator=g)) * alpha
    empty = torch.rand(1, H, W, device=dev, generator=g) < 0.2
    alpha[empty] = 0.0
    depth[empty] = 0.0
    radii = torch.randint(0, 5, (N,), device=dev, dtype=torch.int32, generator=g)
    w = torch.randn(1, H, W, device=dev, generator=g)
    d0, a0 = depth.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ref = torch.nan_to_num(d0 / a0, 0, 0)
    (ref * w).sum().backward()
    d1, a1 = depth.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    out, vis = render_tail(d1, a1, radii)
    (out * w).sum().backward()
    assert torch.equal(vis, radii > 0) and vis.dtype == torch.bool
    assert torch.equal(out, ref)
    ok = ~empty
    assert torch.allclose(d1.grad[ok], d0.grad[ok], rtol=1e-6, atol=0) and torch.allclose(a1.grad[ok], a0.grad[ok], rtol=1e-6, atol=0)
    assert bool((d1.grad[empty] == 0).all()) and bool((a1.grad[empty] == 0).all())   # torch leaves NaN there


@pytest.mark.parametrize("surfels", [False, True])
def test_fused_densify_and_prune_equals_the_sequential_surgery(dev, surfels):
    """One row plan + ONE scorp_gather_rows launch (parameters and both Adam moments, ping-pong arena) against
    densify_and_clone -> densify_and_split -> prune_points done with torch.cat / indexing (the reference's sequence,
    gaussian_model.py:528-584): same row order, same values bit for bit (same random draw), zero moments on the new
    rows, statistics reset; twice in a row, so the second pass runs inside the arena's spare half."""
    import math
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.renderer2d import GaussianModel2D
    from scorp_amd.synthetic import make_gaussians
    cls = GaussianModel2D if surfels else GaussianModel
    raw = make_gaussians(5000, 2, 41, log_scale_mean=math.log(0.05), scale_dims=2 if surfels else 3)
    models = []
    for fused in (False, True):
        m = cls.from_raw(raw, 2, device=dev)
        m.training_setup(OptimizationParams())
        m.fused_densify = fused
        models.append(m)
    g = torch.Generator(device=dev).manual_seed(3)
    names = ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation")
    for rnd in range(2):
        n = models[0].get_xyz.shape[0]
        grads_acc = torch.rand(n, 1, device=dev, generator=g) * 4e-4
        denom = torch.randint(0, 3, (n, 1), device=dev, generator=g).float()          # zeros -> NaN -> 0, as in training
        fake_grads = [torch.randn(getattr(models[0], nm).shape, device=dev, generator=g) for nm in names]
        for m in models:
            for nm, fg in zip(names, fake_grads):   # two optimizer steps: non-trivial Adam moments on every row
                getattr(m, nm).grad = fg.clone()
            m.optimizer.step()
            m.optimizer.zero_grad(set_to_none=True)
            m.xyz_gradient_accum, m.denom = grads_acc.clone(), denom.clone()
            m.max_radii2D = torch.zeros(n, device=dev)
            torch.manual_seed(77 + rnd)
            m.densify_and_prune(2e-4, 0.3, 2.0, 20)
        a, b = models
        assert a.get_xyz.shape[0] == b.get_xyz.shape[0] != n
        for nm in names:
            pa, pb = getattr(a, nm), getattr(b, nm)
            assert torch.equal(pa.detach(), pb.detach()), nm
            sa, sb = a.optimizer.state[pa], b.optimizer.state[pb]
            assert torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"]), nm
            assert pb.requires_grad and pb.is_contiguous()
        assert float(b.denom.sum()) == 0.0 and b.max_radii2D.shape[0] == b.get_xyz.shape[0]


def test_stacked_views_equal_single_views(dev):
    """ScorpGs3dInputs.num_views: V views of one model rendered as ONE stacked image (V x N virtual Gaussians, V x H rows)
    equal the V single renders bit for bit - images, raw depth, alpha, radii - for an SH-3 model; and moving the cameras
    (ViewStack.moved) equals moving an SH-0 object."""
    import copy
    import math
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.multiview import ViewStack, render_stacked
    from scorp_amd.renderer import render
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    from scorp_amd.transforms import gaussians_rotate

    class Pipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False
        fused_activations = True
        raw_outputs = True

    m = GaussianModel.from_raw(make_gaussians(5000, 3, 7, extent=1.0, log_scale_mean=math.log(0.04)), 3, device=dev)
    m.active_sh_degree = 3
    cams = ring_cameras(5, 176, 96, 3, radius=3.0, device=dev)
    bg = torch.tensor([0.3, 0.1, 0.2], device=dev)
    stack = ViewStack(cams, dev)
    out = render_stacked(m, stack, bg)
    assert out["render"].shape == (3, 5 * 96, 176) and out["radii"].shape == (5, 5000)
    with torch.no_grad():
        for v, cam in enumerate(cams):
            one = render(cam, m, Pipe(), bg)
            rows = slice(v * 96, (v + 1) * 96)
            assert torch.equal(out["render"][:, rows], one["render"]), v
            assert torch.equal(out["render_alpha"][rows], one["render_alpha"][0]) and torch.equal(out["render_depth_raw"][rows], one["render_depth_raw"][0])
            assert torch.equal(out["radii"][v], one["radii"])
    # cameras moved the other way == object moved (SH 0: colour does not depend on the direction)
    rots = np.load(os.path.join(os.path.dirname(__file__), "golden", "rotations_128.npz"))["rotations"]
    obj = _object_model(dev, 4000, 0, 5)
    R = torch.tensor(rots[17], dtype=torch.float32, device=dev)
    c = obj._xyz.detach().mean(0)
    view, proj, campos = ViewStack(cams, dev).moved(R, c - c @ R.T)
    a = render_stacked(obj, stack, bg, view, proj, campos)
    moved = copy.copy(obj)
    moved._xyz, moved._rotation = obj._xyz.detach().clone(), obj._rotation.detach().clone()
    gaussians_rotate(moved, R, fix_center=True)
    b = render_stacked(moved, stack, bg)
    assert (a["render"] - b["render"]).abs().mean() < 2e-5 and (a["render_alpha"] - b["render_alpha"]).abs().mean() < 2e-5
    assert (a["render_depth_raw"] - b["render_depth_raw"]).abs().mean() < 1e-4


def test_hip_transform_kernel_matches_wigner_table_and_torch_formulation(dev):
    """scorp_gaussians_transform (one launch: positions, quaternions, log-scales, SH bands 1-3) against (a) the independent
    real Wigner-D table tests/golden/wigner_d.npz applied with numpy and (b) the torch formulation of
    gaussians_rotate / gaussians_scale / gaussians_translate (utils/gaussians.py:12-108), SH-3 3DGS and surfel models."""
    import copy
    import math
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.renderer2d import GaussianModel2D
    from scorp_amd.synthetic import make_gaussians
    from scorp_amd import transforms as TR
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "wigner_d.npz"))
    for k in (0, 3, 7):
        R = torch.tensor(g["rotations"][k], dtype=torch.float32)          # host tensors: the kernel path
        blocks = [torch.tensor(g[n][k]) for n in ("D1", "D2", "D3")]
        for cls, dims in ((GaussianModel, 3), (GaussianModel2D, 2)):
            raw = make_gaussians(3001, 3, 11 + k, scale_dims=dims)
            m = cls.from_raw(raw, 3, device=dev)
            T = torch.tensor([0.3, -0.2, 0.7]); S = torch.tensor([1.5, 1.5, 1.5]) if dims == 3 else None
            TR.gaussians_transform(m, R=R, T=T, scale=S, fix_center=True, blocks=blocks)
            # (a) SH bands by the golden table, in float64 numpy
            rest = raw["features_rest"].astype(np.float64)
            for l, name in ((1, "D1"), (2, "D2"), (3, "D3")):
                sl = slice(l * l - 1, (l + 1) ** 2 - 1)
                rest[:, sl] = np.einsum("ij,njc->nic", g[name][k], rest[:, sl])
            assert np.abs(m._features_rest.detach().cpu().numpy() - rest).max() < 2e-6
            # (b) geometry by the torch formulation on a CPU copy
            ref = cls.from_raw(raw, 3, device="cpu")
            c = ref._xyz.data.mean(0)
            xyz = (ref._xyz.data - c) @ R.T
            if S is not None:
                xyz = xyz * S[None]
            xyz = xyz + c + T
            assert (m._xyz.detach().cpu() - xyz).abs().max() < 2e-6
            q = TR.matrix_to_quat(R)
            rot = TR.quat_multiply(q[None], ref._rotation.data / ref._rotation.data.norm(dim=1, keepdim=True))
            assert (m._rotation.detach().cpu() - rot).abs().max() < 2e-6
            sc = ref._scaling.data + (torch.log(S[:dims])[None] if S is not None else 0.0)
            assert (m._scaling.detach().cpu() - sc).abs().max() < 2e-6
    # the public functions take the kernel for host-side parameters and agree with the torch path (device-side parameters)
    m1 = GaussianModel.from_raw(make_gaussians(2000, 2, 5), 2, device=dev)
    m2 = GaussianModel.from_raw(make_gaussians(2000, 2, 5), 2, device=dev)
    R = torch.tensor(g["rotations"][2], dtype=torch.float32)
    TR.gaussians_rotate(m1, R, fix_center=True)            # kernel
    TR.gaussians_rotate(m2, R.to(dev), fix_center=True)    # torch ops
    for n in ("_xyz", "_rotation", "_features_rest"):
        assert (getattr(m1, n) - getattr(m2, n)).abs().max() < 5e-6, n


@pytest.mark.gpu
def test_view_statistics_kernel_equals_the_masked_indexing(dev):
    """GaussianModel.accumulate_view_stats (one launch of scorp_densification_stats) against the reference's three
    boolean-mask updates (train_3dgs.py:180-181, gaussian_model.py:603-605), with and without the device-side skip word."""
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.synthetic import make_gaussians
    n = 20011
    g = torch.Generator(device="cpu").manual_seed(5)

    class VS:   # what the loop hands over: something with a .grad of shape [N, 3]
        pass

    ours = GaussianModel.from_raw(make_gaussians(n, 1, 3), 1, device=dev)
    ours.training_setup(OptimizationParams())
    ref_max, ref_acc, ref_den = ours.max_radii2D.clone(), ours.xyz_gradient_accum.clone(), ours.denom.clone()
    for view in range(3):
        vs = VS()
        vs.grad = torch.randn(n, 3, generator=g).to(dev) * 1e-3
        radii = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(dev)
        vis = (radii > 0) & (torch.rand(n, generator=g).to(dev) > 0.3)
        skip = torch.tensor([1 if view == 1 else 0], dtype=torch.int32, device=dev)   # the second view "overflowed"
        ours.accumulate_view_stats(vs, vis, radii, skip_flag=skip if view else None)
        if view != 1:
            ref_max[vis] = torch.max(ref_max[vis], radii[vis].float())
            ref_acc[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)
            ref_den[vis] += 1
    torch.cuda.synchronize()
    assert torch.equal(ours.max_radii2D, ref_max) and torch.equal(ours.denom, ref_den)
    torch.testing.assert_close(ours.xyz_gradient_accum, ref_acc, rtol=1e-6, atol=0)
    assert float(ref_den.max()) == 2.0 and float(ref_den.min()) == 0.0


def test_view_statistics_of_the_surfel_model_norm_the_whole_row(dev):
    """The 2DGS model's add_densification_stats takes the norm over all three components of a means2D-gradient row
    (gs2dgs/scene/gaussian_model.py:494-495); the fused update (scorp_densification_stats_ex, norm_components = 3) must
    do the same, and a subclass that overrides add_densification_stats must be served by ITS method."""
    from scorp_amd.gaussian_model import OptimizationParams2D
    from scorp_amd.renderer2d import GaussianModel2D
    from scorp_amd.synthetic import make_gaussians
    n = 5003
    g = torch.Generator(device="cpu").manual_seed(8)

    class VS:
        pass

    class Custom(GaussianModel2D):
        def add_densification_stats(self, viewspace_point_tensor, update_filter):     # anything else: here the L1 norm
            self.xyz_gradient_accum[update_filter] += viewspace_point_tensor.grad[update_filter].abs().sum(-1, keepdim=True)
            self.denom[update_filter] += 1

    raw = make_gaussians(n, 1, 3, scale_dims=2)
    ours, custom = GaussianModel2D.from_raw(raw, 1, device=dev), Custom.from_raw(raw, 1, device=dev)
    for m in (ours, custom):
        m.training_setup(OptimizationParams2D())
    ref_max, ref_acc, ref_den = ours.max_radii2D.clone(), ours.xyz_gradient_accum.clone(), ours.denom.clone()
    ref_l1 = ref_acc.clone()
    for view in range(2):
        vs = VS()
        vs.grad = torch.randn(n, 3, generator=g).to(dev) * 1e-3      # a z component that is NOT zero
        radii = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32).to(dev)
        vis = (radii > 0) & (torch.rand(n, generator=g).to(dev) > 0.3)
        ours.accumulate_view_stats(vs, vis, radii)
        custom.accumulate_view_stats(vs, vis, radii)
        ref_max[vis] = torch.max(ref_max[vis], radii[vis].float())
        ref_acc[vis] += torch.norm(vs.grad[vis], dim=-1, keepdim=True)
        ref_l1[vis] += vs.grad[vis].abs().sum(-1, keepdim=True)
        ref_den[vis] += 1
    torch.cuda.synchronize()
    for m in (ours, custom):
        assert torch.equal(m.max_radii2D, ref_max) and torch.equal(m.denom, ref_den)
    torch.testing.assert_close(ours.xyz_gradient_accum, ref_acc, rtol=1e-6, atol=0)
    torch.testing.assert_close(custom.xyz_gradient_accum, ref_l1, rtol=1e-6, atol=0)


def test_scoring_render_equals_render_then_score_and_is_bit_repeatable(dev):
    import math
    """scorp_gs3d_render_score (the scoring form of the blend: no colour, no images, the comparison with the target in the
    epilogue, per-block partial sums added in a fixed order) against the stacked render followed by
    scorp_gs3d_pose_score_accumulate: the same mismatch per hypothesis band up to float summation order, and the same BITS
    from two runs.  Two hypotheses x three views share the launch set; every band is held against the same target."""
    import ctypes
    from scorp_amd import _C
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.multiview import ViewStack, render_stacked, score_stacked
    from scorp_amd.rasterizer3d import PairPolicy, _stream
    from scorp_amd.synthetic import make_gaussians, ring_cameras
    L = _C.lib()
    m = GaussianModel.from_raw(make_gaussians(4000, 0, 5, log_scale_mean=math.log(0.05)), 0, device=dev)
    cams = ring_cameras(5, 112, 80, 3, radius=3.4, device=dev)
    bg = torch.zeros(3, device=dev)
    PairPolicy.reset()
    try:
        tgt = render_stacked(m, ViewStack(cams[:3], dev), bg)                      # the target: views 0..2
        a_t = tgt["render_alpha"]
        d_t = torch.nan_to_num(tgt["render_depth_raw"] / a_t, 0.0, 0.0).contiguous()
        stack6 = ViewStack([cams[0], cams[1], cams[2], cams[2], cams[3], cams[4]], dev)   # band 0 == the target, band 1 not
        out = render_stacked(m, stack6, bg)
        n = a_t.numel()
        ref = torch.zeros(2, device=dev)
        p = lambda t, off=0: ctypes.c_void_p(t.data_ptr() + 4 * off)
        for j in range(2):
            _C.check(L.scorp_gs3d_pose_score_accumulate(p(out["render_depth_raw"], j * n), p(out["render_alpha"], j * n), p(d_t), p(a_t),
                                                        n, 1.0 / n, ctypes.c_void_p(ref[j:j + 1].data_ptr()), _stream()), "score")
        PairPolicy.mode = "reserve"
        got = []
        for _ in range(2):
            acc = torch.zeros(2, device=dev)
            score_stacked(m, stack6, bg, stack6.view, stack6.proj, stack6.campos, d_t, a_t.contiguous(), acc, 3 * stack6.H, 1.0 / n)
            got.append(acc)
        PairPolicy.drain()
        assert torch.equal(got[0], got[1]), "two scoring renders differ"
        assert float(ref[0]) < 1e-7 and float(ref[1]) > 1e-3, ref            # band 0 IS the target, band 1 is not
        assert float((got[0] - ref).abs().max()) <= 2e-6 * max(float(ref.abs().max()), 1.0), (got[0], ref)
    finally:
        PairPolicy.reset()
