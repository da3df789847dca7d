"""Photometric loss functions with the reference's signatures, on fused gfx950 kernels (csrc/loss.hip).

    from mobgs_amd.loss_utils import l1_loss, ssim, psnr          # drop-in for utils/loss_utils.py, utils/image_utils.py
    loss = photometric_loss(image, gt, lambda_dssim=0.2)          # L1 + lambda * (1 - SSIM) in ONE forward kernel

mirrors /root/reference/utils/loss_utils.py:233-239 (l1_loss), :351-381 (ssim, 11x11 Gaussian window sigma 1.5, zero
padding), /root/reference/utils/image_utils.py:17-38 (psnr) and the combination of /root/reference/train.py:621-628.
Gradients flow to the first argument (the rendered image); the second (ground truth) is treated as a constant, as
in every reference call.  The masked L1 exists in the reference only inside the flow-consistency loss
(train.py:651-671): `flow_warp_loss` below is that whole block -- coordinate normalisation, both grid_sample warps and
both masked L1 terms -- as one forward and one backward kernel (csrc/flowloss.hip), differentiable in all six inputs.
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


class _FlowWarpLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ori, latent, e2m, m2e, la, da, combine_taps):
        lib = _lib.load()
        ori, latent, e2m, m2e, la, da = (f32c(t) for t in (ori, latent, e2m, m2e, la, da))
        B, K, _, H, W = latent.shape
        dev = ori.device
        partial = torch.empty(lib.mobgs_flow_warp_loss_blocks(B, H, W), 4, dtype=torch.float32, device=dev)
        out = torch.empty(5, dtype=torch.float32, device=dev)  # sums[4], loss
        check(lib.mobgs_flow_warp_loss_fwd(B, K, H, W, ptr(ori), ptr(latent), ptr(e2m), ptr(m2e), ptr(la), ptr(da),
                                           ptr(partial), ptr(out), ptr(out[4:]), stream()), "mobgs_flow_warp_loss_fwd")
        ctx.save_for_backward(ori, latent, e2m, m2e, la, da, out)
        ctx.combine_taps = int(combine_taps)
        return out[4]

    @staticmethod
    def backward(ctx, v):
        lib = _lib.load()
        ori, latent, e2m, m2e, la, da, out = ctx.saved_tensors
        B, K, _, H, W = latent.shape
        need = ctx.needs_input_grad
        v = f32c(v).reshape(1)
        g_ori = torch.zeros_like(ori) if need[0] else None          # scatter targets: accumulated with atomics
        g_latent = torch.zeros_like(latent) if need[1] else None
        g_e2m = torch.empty_like(e2m) if need[2] else None
        g_m2e = torch.empty_like(m2e) if need[3] else None
        g_la = torch.empty_like(la) if need[4] else None
        g_da = torch.empty_like(da) if need[5] else None
        scratch = None
        if need[0] or need[1]:
            scratch = torch.empty(lib.mobgs_flow_warp_loss_bwd_scratch_floats(B, K, H, W), dtype=torch.float32,
                                  device=ori.device)
        check(lib.mobgs_flow_warp_loss_bwd(B, K, H, W, ptr(ori), ptr(latent), ptr(e2m), ptr(m2e), ptr(la), ptr(da),
                                           ptr(out), ptr(v), ptr(g_ori), ptr(g_latent), ptr(g_e2m), ptr(g_m2e),
                                           ptr(g_la), ptr(g_da), ptr(scratch), ctx.combine_taps, stream()),
              "mobgs_flow_warp_loss_bwd")
        return g_ori, g_latent, g_e2m, g_m2e, g_la, g_da, None


def flow_warp_loss(ori_image_tensor, latent_img_final_tensor, exp2mid_coord_final_tensor, mid2exp_coord_final_tensor,
                   latent_alpha_final_tensor, d_alpha_tensor, lambda_flow_loss=1.0, combine_taps=True):
    """The flow-consistency term of /root/reference/train.py:651-671 from the tensors of :608-617:

        flow_loss = lambda_flow_loss * (l1(grid_sample(ori, norm(exp2mid)), latent, mask=latent_alpha)
                                        + l1(grid_sample(latent, norm(mid2exp)), ori, mask=d_alpha))

    ori [B,3,H,W]; latent [B,K,3,H,W]; exp2mid / mid2exp [B,K,H,W,2] pixel coordinates exactly as get_flow() returns
    them (the reference normalises them in place before sampling; this function does not modify its arguments);
    latent_alpha [B,K,1,H,W] (or [B,K,H,W]); d_alpha [B,1,H,W] (or [B,H,W]).  Differentiable in all six tensors.
    lambda_flow_loss == 0 (the shipped seesaw / children configs): a zero that is part of no graph -- the reference
    evaluates and back-propagates the whole block to multiply it by zero."""
    B, K = latent_img_final_tensor.shape[:2]
    H, W = ori_image_tensor.shape[-2:]
    if latent_img_final_tensor.shape != (B, K, 3, H, W) or ori_image_tensor.shape != (B, 3, H, W):
        raise ValueError("flow_warp_loss: ori [B,3,H,W] and latent [B,K,3,H,W] expected")
    for name, t in (("exp2mid", exp2mid_coord_final_tensor), ("mid2exp", mid2exp_coord_final_tensor)):
        if t.shape != (B, K, H, W, 2):
            raise ValueError(f"flow_warp_loss: {name} coordinates must be [B,K,H,W,2], got {tuple(t.shape)}")
    if latent_alpha_final_tensor.numel() != B * K * H * W or d_alpha_tensor.numel() != B * H * W:
        raise ValueError("flow_warp_loss: latent_alpha [B,K,1,H,W] and d_alpha [B,1,H,W] expected")
    if isinstance(lambda_flow_loss, (int, float)) and lambda_flow_loss == 0:
        return ori_image_tensor.new_zeros(())
    loss = _FlowWarpLoss.apply(ori_image_tensor, latent_img_final_tensor, exp2mid_coord_final_tensor,
                               mid2exp_coord_final_tensor, latent_alpha_final_tensor.reshape(B, K, H, W),
                               d_alpha_tensor.reshape(B, H, W), combine_taps)
    return lambda_flow_loss * loss


@torch.no_grad()
def psnr(img1, img2, mask=None):
    """/root/reference/utils/image_utils.py:17-38 (mask=None branch): per-image 20 log10(1 / sqrt(mse))."""
    if mask is not None:
        raise NotImplementedError("masked psnr is not on the training path")
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse.float()))
