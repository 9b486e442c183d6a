"""Host logic of the Gaussian model that needs no GPU."""
import torch


def test_view_statistics_without_mask_indexing_equal_the_reference_form():
    """accumulate_view_stats on CPU tensors (the torch.where form) against train_3dgs.py:180-181 written as in the
    reference: same max_radii2D, denom and gradient accumulator, also behind a skip word."""
    from scorp_amd.gaussian_model import GaussianModel, OptimizationParams
    from scorp_amd.synthetic import make_gaussians
    n = 777
    g = torch.Generator().manual_seed(3)
    m = GaussianModel.from_raw(make_gaussians(n, 1, 3), 1, device="cpu")
    m.training_setup(OptimizationParams())
    ref_max, ref_acc, ref_den = m.max_radii2D.clone(), m.xyz_gradient_accum.clone(), m.denom.clone()

    class VS:
        pass

    for view in range(3):
        vs = VS()
        vs.grad = torch.randn(n, 3, generator=g) * 1e-3
        radii = torch.randint(0, 40, (n,), generator=g, dtype=torch.int32)
        vis = (radii > 0) & (torch.rand(n, generator=g) > 0.3)
        skip = torch.tensor([1 if view == 1 else 0], dtype=torch.int32)
        m.accumulate_view_stats(vs, vis, radii, skip_flag=skip if view else None)
        if view != 1:
            ref_max[vis] = torch.max(ref_max[vis], radii[vis].float())
            ref_acc[vis] += torch.norm(vs.grad[vis, :2], dim=-1, keepdim=True)
            ref_den[vis] += 1
    assert torch.equal(m.max_radii2D, ref_max) and torch.equal(m.denom, ref_den)
    torch.testing.assert_close(m.xyz_gradient_accum, ref_acc, rtol=1e-6, atol=0)


def test_fused_adam_rollback_takes_discarded_steps_back():
    """FusedAdam counts a bias-correction step on the host for every view, the ones the device discarded included (a fused view
    that overflowed its pair reservation); rollback_steps(n) takes them out again - for the parameters the last step() advanced
    only, and never below zero.  (Host bookkeeping: no GPU involved; the GPU side is
    tests/test_train_gpu.py::test_first_view_overflow_retry_equals_a_presized_run.)"""
    import torch
    from scorp_amd.fused_adam import FusedAdam
    from scorp_amd.rasterizer3d import PairOverflow
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
    opt = FusedAdam([{"params": [a]}, {"params": [b]}], lr=1e-3)
    opt.state[a]["step"], opt.state[b]["step"] = torch.tensor(5.0), torch.tensor(7.0)
    opt._stepped = [opt.state[a]]                 # the last step() advanced `a` only (`b` frozen: no gradient)
    opt.rollback_steps(2)
    assert float(opt.state[a]["step"]) == 3.0 and float(opt.state[b]["step"]) == 7.0
    opt.rollback_steps(9)
    assert float(opt.state[a]["step"]) == 0.0
    opt.rollback_steps(0)
    e = PairOverflow("pair reservation too small", 3)
    assert isinstance(e, RuntimeError) and e.count == 3


def test_bench_self_launcher_builds_the_torchrun_command(monkeypatch):
    """`python bench.py --gpus N` from a bare shell starts its ranks as CHILD processes of torch.distributed.run (never an exec,
    never a GPU call in the parent) with the same arguments and relays the return code."""
    import subprocess
    import sys
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.launch_ranks(4) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    assert cmd[-7].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_add_densification_stats_without_mask_compaction_equals_the_reference_statements():
    """gaussian_model.py:603-605 (3DGS: norm over x, y) and gs2dgs/scene/gaussian_model.py:494-495 (2DGS: the whole row): with a
    boolean mask the update runs over all rows - same bits as the two boolean-mask indexings - and an index tensor takes the
    reference's statements as they stand."""
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.renderer2d import GaussianModel2D
    from scorp_amd.synthetic import make_gaussians
    torch.manual_seed(3)
    for cls, dims in ((GaussianModel, 3), (GaussianModel2D, 2)):
        raw = make_gaussians(500, 1, 1, scale_dims=dims)
        m = cls.from_raw(raw, 1, device="cpu")
        a0, d0 = torch.rand(500, 1), torch.rand(500, 1).round()
        vp = torch.zeros(500, 3, requires_grad=True)
        vp.grad = torch.randn(500, 3)
        f = torch.rand(500) > 0.4
        cols = vp.grad[f, :2] if dims == 3 else vp.grad[f]
        ra, rd = a0.clone(), d0.clone()
        ra[f] += torch.norm(cols, dim=-1, keepdim=True)
        rd[f] += 1
        for filt in (f, torch.nonzero(f).squeeze(-1)):
            m.xyz_gradient_accum, m.denom = a0.clone(), d0.clone()
            m.add_densification_stats(vp, filt)
            assert torch.equal(m.xyz_gradient_accum, ra) and torch.equal(m.denom, rd)


def test_render_takes_raw_leaves_by_itself_only_for_a_stock_model():
    """renderer._fused_activations: no `fused_activations` attribute on the pipe (the reference's PipelineParams) -> the raw-leaf
    path iff the model's activation functions and getters are this package's own; an explicit flag decides otherwise."""
    from scorp_amd.gaussian_model import GaussianModel
    from scorp_amd.renderer import _fused_activations
    from scorp_amd.renderer2d import GaussianModel2D, _fused_activations as fused2d
    from scorp_amd.synthetic import make_gaussians

    class RefPipe:
        convert_SHs_python = False
        compute_cov3D_python = False
        debug = False

    class Off(RefPipe):
        fused_activations = False

    class On(RefPipe):
        fused_activations = True

    class Doubled(GaussianModel):
        @property
        def get_scaling(self):
            return 2.0 * super().get_scaling

    m = GaussianModel.from_raw(make_gaussians(10, 1, 1), 1, device="cpu")
    m2 = GaussianModel2D.from_raw(make_gaussians(10, 1, 1, scale_dims=2), 1, device="cpu")
    sub = Doubled.from_raw(make_gaussians(10, 1, 1), 1, device="cpu")
    assert _fused_activations(RefPipe, m) and fused2d(RefPipe, m2)
    assert not _fused_activations(RefPipe, sub)              # an overridden getter: the torch activations are what it means
    assert not _fused_activations(Off, m) and _fused_activations(On, sub)
    m.opacity_activation = lambda x: torch.sigmoid(x) * 0.5
    assert not _fused_activations(RefPipe, m)
    assert not _fused_activations(RefPipe, object())         # (somebody else's model class)


def test_host_patch_keeps_the_function_object_and_falls_back_to_the_original(monkeypatch):
    """scorp_amd.hostpatch.patch_ssim on a stand-in for gs3dgs/utils/loss_utils.py: names bound BEFORE the patch see the new body,
    CPU tensors (and anything the HIP kernels do not serve) take the original code bit for bit, a second patch is a no-op, and
    nothing happens without SCORP_AMD_ACCELERATE=1."""
    import sys
    import types
    from scorp_amd import hostpatch
    from scorp_amd.loss import ssim_torch
    src = '''
import torch
import torch.nn.functional as F
from math import exp
def gaussian(window_size, sigma):
    gauss = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return gauss / gauss.sum()
def create_window(window_size, channel):
    w = gaussian(window_size, 1.5).unsqueeze(1)
    return w.mm(w.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()
def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = create_window(window_size, channel).type_as(img1)
    mu1 = F.conv2d(img1, window, padding=window_size // 2, groups=channel)
    mu2 = F.conv2d(img2, window, padding=window_size // 2, groups=channel)
    s1 = F.conv2d(img1 * img1, window, padding=window_size // 2, groups=channel) - mu1 * mu1
    s2 = F.conv2d(img2 * img2, window, padding=window_size // 2, groups=channel) - mu2 * mu2
    s12 = F.conv2d(img1 * img2, window, padding=window_size // 2, groups=channel) - mu1 * mu2
    m = ((2 * mu1 * mu2 + 1e-4) * (2 * s12 + 9e-4)) / ((mu1 * mu1 + mu2 * mu2 + 1e-4) * (s1 + s2 + 9e-4))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)
'''
    mod = types.ModuleType("gs3dgs.utils.loss_utils")
    exec(compile(src, "loss_utils_standin.py", "exec"), mod.__dict__)
    early = mod.ssim                                   # what `from gs3dgs.utils.loss_utils import ssim` bound earlier
    x, y = torch.rand(3, 40, 50), torch.rand(3, 40, 50)
    before = early(x, y)
    monkeypatch.setitem(sys.modules, "gs3dgs.utils.loss_utils", mod)
    monkeypatch.delenv("SCORP_AMD_ACCELERATE", raising=False)
    assert hostpatch.accelerate_reference() == [] and not getattr(mod.ssim, "_scorp_patched", False)
    monkeypatch.setenv("SCORP_AMD_ACCELERATE", "1")
    assert hostpatch.accelerate_reference() == ["gs3dgs.utils.loss_utils"]
    assert mod.ssim is early and early._scorp_patched and hostpatch.accelerate_reference() == []
    assert torch.equal(early(x, y), before) and torch.equal(mod.ssim_torch(x, y), before)      # CPU: the original code
    assert torch.equal(early(x[None], y[None], 7, False), mod.ssim_torch(x[None], y[None], 7, False))
    assert abs(float(before) - float(ssim_torch(x, y))) < 1e-6
