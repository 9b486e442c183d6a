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
    from scorp_amd.loss import l1_loss, ssim
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
