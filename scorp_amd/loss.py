"""Photometric losses of the training step — mirrors gs3dgs/utils/loss_utils.py:17-73 and
gs3dgs/utils/image_utils.py:18-20: `l1_loss`, `ssim` (11x11 Gaussian window, sigma 1.5, zero padding 5,
C1=0.01^2, C2=0.03^2, mean over all elements) and `psnr`.

`ssim_torch` is the torch formulation (five depthwise convolutions, as in the reference) with the window cached per
device instead of rebuilt on the CPU and uploaded on every call (loss_utils.py:45-49).  `ssim` - the reference's name,
what an unmodified training script calls - answers from the HIP loss kernels (scorp_amd.fused_loss, lambda = 1) when its
arguments are what train_3dgs.py:106-107 passes (a C x H x W float32 GPU image against a ground truth that needs no
gradient, the default 11-wide window, the scalar mean) and from `ssim_torch` otherwise: MIOpen's depthwise convolutions
take 9.4 ms per 1600x1200 view, forward + backward (85 % of the reference's call pattern around this library's
rasterizer), the HIP kernels 0.1 ms.
"""
from math import exp

import torch
import torch.nn.functional as F

_WINDOWS = {}


def l1_loss(network_output, gt):
    return torch.abs(network_output - gt).mean()


def l2_loss(network_output, gt):
    return ((network_output - gt) ** 2).mean()


def gaussian(window_size, sigma):
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    return g / g.sum()


def create_window(window_size, channel):
    w1 = gaussian(window_size, 1.5).unsqueeze(1)
    w2 = w1.mm(w1.t()).float().unsqueeze(0).unsqueeze(0)
    return w2.expand(channel, 1, window_size, window_size).contiguous()


def _window(window_size, channel, like):
    key = (window_size, channel, like.device, like.dtype)
    if key not in _WINDOWS:
        _WINDOWS[key] = create_window(window_size, channel).to(device=like.device, dtype=like.dtype)
    return _WINDOWS[key]


def _hip_ssim_applies(img1, img2, window_size, size_average):
    return (torch.is_tensor(img1) and torch.is_tensor(img2) and img1.is_cuda and img2.is_cuda and img1.dim() == 3
            and img1.shape == img2.shape and img1.dtype == torch.float32 and img2.dtype == torch.float32
            and window_size == 11 and size_average and not img2.requires_grad)


def ssim(img1, img2, window_size=11, size_average=True):
    """gs3dgs/utils/loss_utils.py:51-73 by name and value; see the module docstring for which arguments the HIP kernels
    serve.  (The kernels return (1 - lambda) L1 + lambda (1 - SSIM); lambda = 1 leaves 1 - SSIM.)"""
    if _hip_ssim_applies(img1, img2, window_size, size_average):
        from .fused_loss import fused_l1_ssim_loss
        return 1.0 - fused_l1_ssim_loss(img1, img2, 1.0)
    return ssim_torch(img1, img2, window_size, size_average)


def ssim_torch(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _window(window_size, channel, img1)
    pad = window_size // 2
    conv = lambda x: F.conv2d(x, window, padding=pad, groups=channel)
    mu1, mu2 = conv(img1), conv(img2)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = conv(img1 * img1) - mu1_sq
    sigma2_sq = conv(img2 * img2) - mu2_sq
    sigma12 = conv(img1 * img2) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def photometric_loss(image, gt, lambda_dssim=0.2):
    """(1-lambda) L1 + lambda (1-SSIM), train_3dgs.py:106-107."""
    return (1.0 - lambda_dssim) * l1_loss(image, gt) + lambda_dssim * (1.0 - ssim(image, gt))


def isotropic_loss(scaling):
    """Mean |s - mean_axis(s)| over all Gaussians and axes (gs3dgs/utils/loss_utils.py:75-85)."""
    return torch.abs(scaling - scaling.mean(dim=1, keepdim=True)).mean()


def depth_normalize_(depth):
    """Min-max normalisation with detached extrema (gs3dgs/utils/image_utils.py:87-91)."""
    min_val, max_val = torch.min(depth).detach(), torch.max(depth).detach()
    return (depth - min_val) / (max_val - min_val)


def depth_losses(rend_depth, iteration, opt, gt_depth=None, gt_depth_est=None):
    """The depth terms of train_3dgs.py:109-133 / train_2dgs.py:95-133 (active after opt.depth_from_iter): L1 against a
    sensor depth inside (0.3, 7) where something was rendered, and L1 between min-max-normalised rendered and estimated
    (monocular) depth with the exponentially decaying weight 10 * dn_l1_weight(iteration)."""
    from .gaussian_model import get_expon_lr_func
    loss = torch.zeros((), device=rend_depth.device)
    if iteration <= opt.depth_from_iter:
        return loss
    if gt_depth is not None:
        mask = (gt_depth > 0.3) & (gt_depth < 7) & (rend_depth > 0.0)
        loss = loss + opt.lambda_depth_sensor * l1_loss(rend_depth[mask], gt_depth[mask])
    if gt_depth_est is not None:
        w = get_expon_lr_func(opt.dn_l1_weight_init, opt.dn_l1_weight_final, max_steps=opt.iterations)(iteration)
        mask = (rend_depth > 0.0) & (gt_depth_est > 0.0)
        loss = loss + 10 * w * l1_loss(depth_normalize_(rend_depth[mask]), depth_normalize_(gt_depth_est[mask]))
    return loss
