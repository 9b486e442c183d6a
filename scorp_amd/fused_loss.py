"""Fused L1 + SSIM training loss on the HIP library (scorp_amd/csrc/loss.hip).

`fused_l1_ssim_loss(image, gt, lambda_dssim, mask)` equals
`(1-lambda) * l1_loss(image*mask, gt*mask) + lambda * (1 - ssim(image*mask, gt*mask))` of scorp_amd.loss /
gs3dgs/utils/loss_utils.py, differentiable w.r.t. `image`, in two kernels instead of ~10 convolutions.
"""
import ctypes

import torch

from . import _C


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(_C.current_stream_ptr())


class _FusedL1SSIM(torch.autograd.Function):
    @staticmethod
    def forward(ctx, image, gt, lambda_dssim, mask):
        L = _C.lib()
        if not image.is_cuda:
            raise RuntimeError("fused_l1_ssim_loss needs GPU tensors (scorp_amd has no CPU path)")
        image = image.contiguous().float()
        gt = gt.contiguous().float()
        if mask is not None:
            mask = mask.expand(1, *image.shape[-2:]).contiguous().float()
        C, H, W = image.shape
        ws_bytes = L.scorp_loss_workspace_bytes(C, H, W)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=image.device)
        out = torch.empty(3, dtype=torch.float32, device=image.device)
        need_bwd = 1 if ctx.needs_input_grad[0] else 0
        _C.check(L.scorp_loss_l1_ssim_forward(_p(image), _p(gt), _p(mask), C, H, W, float(lambda_dssim), _p(out), _p(ws),
                                              ws_bytes, need_bwd, _stream()), "scorp_loss_l1_ssim_forward")
        ctx.lambda_dssim = float(lambda_dssim)
        ctx.has_mask = mask is not None
        ctx.save_for_backward(image, gt, mask if mask is not None else torch.empty(0, device=image.device), ws)
        ctx.parts = out
        return out[0]      # a view of the three-float result: no copy kernel

    @staticmethod
    def backward(ctx, grad_out):
        L = _C.lib()
        image, gt, mask, ws = ctx.saved_tensors
        mask = mask if ctx.has_mask else None
        C, H, W = image.shape
        grad = torch.empty_like(image)
        go = grad_out.contiguous().float().reshape(1)
        _C.check(L.scorp_loss_l1_ssim_backward(_p(image), _p(gt), _p(mask), C, H, W, ctx.lambda_dssim, _p(ws), _p(go),
                                               _p(grad), _stream()), "scorp_loss_l1_ssim_backward")
        return grad, None, None, None


def fused_l1_ssim_loss(image, gt, lambda_dssim=0.2, mask=None):
    return _FusedL1SSIM.apply(image, gt, lambda_dssim, mask)
