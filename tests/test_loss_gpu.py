"""GPU parity of the fused L1+SSIM loss: against the golden values captured from the reference's own
loss_utils (tests/golden/ref_helpers.npz, G3) and against the torch formulation on ragged / masked inputs."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_fused_loss_matches_reference_golden(golden, dev):
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    a = torch.tensor(golden["g3_a"], device=dev, requires_grad=True)
    b = torch.tensor(golden["g3_b"], device=dev)
    loss = fused_l1_ssim_loss(a, b, 0.2)
    loss.backward()
    assert abs(loss.item() - float(golden["g3_loss"])) < 2e-6
    np.testing.assert_allclose(a.grad.cpu().numpy(), golden["g3_grad_a"], atol=2e-9, rtol=2e-3)


@pytest.mark.parametrize("shape,masked,lam", [((3, 97, 131), False, 0.2), ((3, 64, 64), True, 0.2), ((1, 33, 200), False, 0.5),
                                             ((3, 1200, 1600), False, 0.2), ((3, 5, 7), True, 0.8)])
def test_fused_loss_matches_torch_formulation(shape, masked, lam, dev):
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.loss import l1_loss, ssim_torch as ssim
    g = torch.Generator(device=dev).manual_seed(shape[1])
    x = torch.rand(shape, device=dev, generator=g)
    y = (x + 0.1 * torch.randn(shape, device=dev, generator=g)).clamp(0, 1)
    y[:, : shape[1] // 2, : shape[2] // 3] = x[:, : shape[1] // 2, : shape[2] // 3]      # exact-equal region: sign(0) = 0
    mask = (torch.rand((1,) + shape[1:], device=dev, generator=g) > 0.3).float() if masked else None
    up = torch.tensor(1.7, device=dev)
    x1 = x.clone().requires_grad_(True)
    (fused_l1_ssim_loss(x1, y, lam, mask) * up).backward()
    x2 = x.clone().requires_grad_(True)
    xm, ym = (x2 * mask, y * mask) if masked else (x2, y)
    ref = (1 - lam) * l1_loss(xm, ym) + lam * (1 - ssim(xm, ym))
    (ref * up).backward()
    got = fused_l1_ssim_loss(x, y, lam, mask)
    assert abs(got.item() - ref.item()) < 5e-6
    scale = x2.grad.abs().max().item()
    assert (x1.grad - x2.grad).abs().max().item() < 2e-3 * scale


@pytest.mark.parametrize("shape,box", [((3, 200, 300), (60, 120, 100, 180)), ((3, 1200, 1600), (500, 700, 640, 980)),
                                       ((1, 97, 131), (0, 9, 120, 131)), ((3, 64, 200), (0, 0, 0, 0))])
def test_masked_loss_over_mostly_empty_mask_takes_the_same_values(shape, box, dev):
    """A mask that is one object's silhouette (post_refine_gs.py:103-111) leaves most strips / tiles of the image empty; the
    kernels answer those from constants.  Against the torch formulation, and BIT FOR BIT against the same kernels run the
    long way round (no mask, inputs multiplied by it beforehand): the loss value, and the gradient = unmasked gradient x mask."""
    from scorp_amd.fused_loss import fused_l1_ssim_loss
    from scorp_amd.loss import l1_loss, ssim_torch as ssim
    lam = 0.2
    g = torch.Generator(device=dev).manual_seed(shape[2])
    x = torch.rand(shape, device=dev, generator=g)
    y = (x + 0.1 * torch.randn(shape, device=dev, generator=g)).clamp(0, 1)
    mask = torch.zeros((1,) + shape[1:], device=dev)
    mask[:, box[0]:box[1], box[2]:box[3]] = (torch.rand((1, box[1] - box[0], box[3] - box[2]), device=dev, generator=g) > 0.1).float()
    x1 = x.clone().requires_grad_(True)
    l_masked = fused_l1_ssim_loss(x1, y, lam, mask)
    l_masked.backward()
    xm = (x * mask).requires_grad_(True)
    l_plain = fused_l1_ssim_loss(xm, y * mask, lam)
    l_plain.backward()
    assert l_masked.item() == l_plain.item()
    assert torch.equal(x1.grad, xm.grad * mask)
    x2 = x.clone().requires_grad_(True)
    ref = (1 - lam) * l1_loss(x2 * mask, y * mask) + lam * (1 - ssim(x2 * mask, y * mask))
    ref.backward()
    assert abs(l_masked.item() - ref.item()) < 5e-6
    assert (x1.grad - x2.grad).abs().max().item() <= 2e-3 * max(x2.grad.abs().max().item(), 1e-12)


def test_ssim_by_the_reference_name_is_served_by_the_hip_kernels(dev):
    """`scorp_amd.loss.ssim(image, gt)` as train_3dgs.py:107 calls it: value and gradient of the torch formulation, from the HIP
    loss kernels; other argument forms (a window size, per-image means, a batch dimension, CPU tensors) take the torch form."""
    from scorp_amd.loss import l1_loss, ssim, ssim_torch
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.rand((3, 120, 200), device=dev, generator=g)
    y = (x + 0.1 * torch.randn((3, 120, 200), device=dev, generator=g)).clamp(0, 1)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a = 0.8 * l1_loss(x1, y) + 0.2 * (1.0 - ssim(x1, y))
    b = 0.8 * l1_loss(x2, y) + 0.2 * (1.0 - ssim_torch(x2, y))
    assert type(ssim(x1, y).grad_fn).__name__ != type(ssim_torch(x2, y).grad_fn).__name__   # (not the same graph)
    a.backward(); b.backward()
    assert abs(a.item() - b.item()) < 5e-6
    assert (x1.grad - x2.grad).abs().max().item() < 2e-3 * x2.grad.abs().max().item()
    assert abs(ssim(x, y, 7).item() - ssim_torch(x, y, 7).item()) == 0.0                       # torch form
    assert torch.equal(ssim(x[None], y[None], 11, False), ssim_torch(x[None], y[None], 11, False))
    assert abs(ssim(x.cpu(), y.cpu()).item() - ssim_torch(x.cpu(), y.cpu()).item()) == 0.0


def test_host_patched_reference_ssim_runs_on_the_hip_kernels(dev):
    """scorp_amd.hostpatch.patch_ssim (SCORP_AMD_ACCELERATE=1): a module-level torch `ssim` - here a clone of the torch formulation
    in a module of its own, standing in for gs3dgs/utils/loss_utils.py - keeps its function object and answers GPU calls of the
    training shape from the HIP loss kernels: same value and gradient as its original code (kept as `ssim_torch`)."""
    import inspect
    import types
    import scorp_amd.loss as L
    from scorp_amd.hostpatch import patch_ssim
    mod = types.ModuleType("standin.loss_utils")
    exec(compile(inspect.getsource(L), "standin_loss_utils.py", "exec"), mod.__dict__)
    mod.ssim = types.FunctionType(mod.ssim_torch.__code__, mod.__dict__, "ssim", mod.ssim_torch.__defaults__)
    early = mod.ssim
    assert patch_ssim(mod) and mod.ssim is early and not patch_ssim(mod)
    g = torch.Generator(device=dev).manual_seed(11)
    x = torch.rand((3, 150, 170), device=dev, generator=g)
    y = (x + 0.1 * torch.randn((3, 150, 170), device=dev, generator=g)).clamp(0, 1)
    x1, x2 = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
    a, b = early(x1, y), mod.ssim_torch(x2, y)
    assert any("FusedL1SSIM" in type(f[0]).__name__ for f in a.grad_fn.next_functions if f[0] is not None)   # (1 - fused loss)
    a.backward(); b.backward()
    assert abs(a.item() - b.item()) < 5e-6
    assert (x1.grad - x2.grad).abs().max().item() < 2e-3 * x2.grad.abs().max().item()
    assert torch.equal(early(x, y, 7), mod.ssim_torch(x, y, 7))            # other arguments: the original code
