"""CPU ORACLE (test infrastructure, NOT product code) -- plain-PyTorch restatement of the reference's photometric
loss functions:
    l1_loss   /root/reference/utils/loss_utils.py:233-239
    ssim      /root/reference/utils/loss_utils.py:251-260, 351-381
    psnr      /root/reference/utils/image_utils.py:17-38
PINNED by tests/golden/losses.npz, produced by running the reference's own functions (tests/golden/make_golden.py
gen_losses).  Only tests/ and scripts/bench_modes.py (as the "reference call pattern" timing leg) import this."""
from math import exp

import torch
import torch.nn.functional as F


def l1_loss(a, b):
    return torch.abs(a - b).mean()


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    g = torch.Tensor([exp(-(x - window_size // 2) ** 2 / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    window = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0).expand(channel, 1, window_size, window_size).contiguous()
    window = window.to(img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    s1 = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    s2 = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    s12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1_mu2 + C1) * (2 * s12 + C2)) / ((mu1_sq + mu2_sq + C1) * (s1 + s2 + C2))
    return m.mean() if size_average else m.mean(1).mean(1).mean(1)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).reshape(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse.float()))
