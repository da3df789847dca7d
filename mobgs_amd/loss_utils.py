"""Photometric loss functions with the reference's signatures, on fused gfx950 kernels (csrc/loss.hip).

    from mobgs_amd.loss_utils import l1_loss, ssim, psnr          # drop-in for utils/loss_utils.py, utils/image_utils.py
    loss = photometric_loss(image, gt, lambda_dssim=0.2)          # L1 + lambda * (1 - SSIM) in ONE forward kernel

mirrors /root/reference/utils/loss_utils.py:233-239 (l1_loss), :351-381 (ssim, 11x11 Gaussian window sigma 1.5, zero
padding), /root/reference/utils/image_utils.py:17-38 (psnr) and the combination of /root/reference/train.py:621-628.
Gradients flow to the first argument (the rendered image); the second (ground truth) is treated as a constant, as
in every reference call.  Masked L1 (used only by the flow loss, weight 0 in the shipped configs) stays in torch.
"""
from __future__ import annotations

import torch

from . import _lib
from ._lib import check, f32c, ptr, stream


class _SsimL1(torch.autograd.Function):
    """(img1, img2) [C,H,W] -> per-channel sums of the SSIM map and of |img1 - img2|."""

    @staticmethod
    def forward(ctx, img1, img2):
        lib = _lib.load()
        img1, img2 = f32c(img1), f32c(img2)
        C, H, W = img1.shape
        dev = img1.device
        nb = lib.mobgs_ssim_l1_blocks(C, H, W)
        partial = torch.empty(nb, 2, dtype=torch.float32, device=dev)
        need_grad = ctx.needs_input_grad[0]
        dmaps = torch.empty(3, C, H, W, dtype=torch.float32, device=dev) if need_grad else None
        check(lib.mobgs_ssim_l1_fwd(C, H, W, ptr(img1), ptr(img2), ptr(partial), ptr(dmaps), stream()),
              "mobgs_ssim_l1_fwd")
        sums = partial.reshape(C, -1, 2).sum(dim=1)  # [C,2], fixed order -> deterministic
        ctx.save_for_backward(img1, img2, dmaps)
        return sums[:, 0], sums[:, 1]

    @staticmethod
    def backward(ctx, v_ssim_sum, v_l1_sum):
        lib = _lib.load()
        img1, img2, dmaps = ctx.saved_tensors
        C, H, W = img1.shape
        zero = torch.zeros(C, dtype=torch.float32, device=img1.device)
        scales = torch.stack([f32c(v_ssim_sum) if v_ssim_sum is not None else zero,
                              f32c(v_l1_sum) if v_l1_sum is not None else zero], dim=1).contiguous()
        v_img1 = torch.empty_like(img1)
        check(lib.mobgs_ssim_l1_bwd(C, H, W, ptr(img1), ptr(img2), ptr(dmaps), ptr(scales), ptr(v_img1), stream()),
              "mobgs_ssim_l1_bwd")
        return v_img1, None


def _sums(img1, img2):
    if img1.shape != img2.shape:
        raise ValueError("image shapes differ")
    if img2.requires_grad:
        raise NotImplementedError("mobgs_amd.loss_utils: only the first image receives a gradient")
    H, W = img1.shape[-2:]
    s, l1 = _SsimL1.apply(img1.reshape(-1, H, W), img2.reshape(-1, H, W))
    return s, l1, H * W


def l1_loss(network_output, gt, mask=None):
    if mask is not None:
        channel = gt.shape[1]
        mask = mask.expand(-1, channel, -1, -1)
        return torch.abs((network_output - gt) * mask).sum() / (mask.sum() + 1e-8)
    # the reference's l1_loss is a plain abs().mean() that differentiates both arguments and takes any shape
    # (utils/loss_utils.py:233-239); the fused SSIM+L1 kernel is for image pairs whose second member is a constant.
    # An L1 alone (e.g. the depth term, train.py:651) does not need the 11x11 window pass either: torch.
    return torch.abs(network_output - gt).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    if window_size != 11:
        raise NotImplementedError("the fused SSIM kernel is built for window_size = 11 (the reference's default)")
    s, _, hw = _sums(img1, img2)
    if size_average:
        return s.sum() / (s.numel() * hw)
    if img1.dim() != 4:
        raise ValueError("size_average=False needs [B,C,H,W] input")
    B, C = img1.shape[:2]
    return s.reshape(B, C).sum(1) / (C * hw)


def photometric_loss(image, gt, lambda_dssim=0.2):
    """L1 + lambda_dssim * (1 - SSIM) (train.py:621-628) from one forward and one backward kernel."""
    s, l1, hw = _sums(image, gt)
    n = s.numel() * hw
    ll1 = l1.sum() / n
    if lambda_dssim == 0:
        return ll1
    return ll1 + lambda_dssim * (1.0 - s.sum() / n)


@torch.no_grad()
def psnr(img1, img2, mask=None):
    """/root/reference/utils/image_utils.py:17-38 (mask=None branch): per-image 20 log10(1 / sqrt(mse))."""
    if mask is not None:
        raise NotImplementedError("masked psnr is not on the training path")
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse.float()))
