"""oracle.render_torch.flow_warp_loss (the restatement of train.py:651-671) against closed forms: pins the coordinate
convention (normalised with size - 1, sampled with align_corners=False, border clamp) and the masked-L1 normalisation
that the HIP kernel (csrc/flowloss.hip) is then compared with in tests/test_gpu_flow_loss.py."""
import numpy as np
import torch

from oracle import render_torch as RT


def _pix(B, K, H, W):
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    return torch.stack([xs, ys], dim=-1).expand(B, K, H, W, 2).contiguous()


def test_constant_images_give_twice_their_distance():
    B, K, H, W = 2, 3, 7, 9
    g = torch.Generator().manual_seed(0)
    ori = torch.full((B, 3, H, W), 0.7)
    latent = torch.full((B, K, 3, H, W), 0.2)
    coords = _pix(B, K, H, W) + 3 * torch.randn(B, K, H, W, 2, generator=g)
    la, da = torch.rand(B, K, 1, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)
    loss = RT.flow_warp_loss(ori, latent, coords, coords.flip(0), la, da)
    assert abs(float(loss) - 2 * 0.5) < 1e-5


def test_ramp_image_reads_back_the_sample_position():
    # ori = x (a ramp): the bilinear sample at pixel coordinate c is ix = clip(c W / (W - 1) - 0.5, 0, W - 1) -- the
    # reference normalises with W - 1 but samples with align_corners=False, so c = x does NOT read pixel x back
    B, K, H, W = 1, 2, 5, 11
    ramp = torch.arange(W, dtype=torch.float32).expand(B, 3, H, W).contiguous()
    latent = torch.zeros(B, K, 3, H, W)
    g = torch.Generator().manual_seed(1)
    c = _pix(B, K, H, W) + 4 * torch.randn(B, K, H, W, 2, generator=g)
    la = torch.ones(B, K, 1, H, W)
    da = torch.zeros(B, 1, H, W)  # second term: numerator 0, denominator 1e-8
    loss = RT.flow_warp_loss(ramp, latent, c, c, la, da)
    ix = np.clip(c[..., 0].double().numpy() * W / (W - 1) - 0.5, 0, W - 1)
    assert abs(float(loss) - ix.mean()) < 1e-4 * ix.mean()


def test_mask_weights_the_mean():
    B, K, H, W = 1, 1, 4, 6
    ori = torch.zeros(B, 3, H, W)
    latent = torch.ones(B, K, 3, H, W)
    latent[..., :, :3] = 3.0
    coords = _pix(B, K, H, W)
    la = torch.zeros(B, K, 1, H, W)
    la[..., :3] = 0.5     # only the left half counts: |0 - 3| * 0.5 summed / (0.5 summed)
    loss = RT.flow_warp_loss(ori, latent, coords, coords, la, torch.zeros(B, 1, H, W))
    assert abs(float(loss) - 3.0) < 1e-5
