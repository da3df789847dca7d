"""Fused L1 + SSIM kernels against the fixture produced by the reference's utils/loss_utils.py / image_utils.py."""
import numpy as np
import pytest
import torch

from helpers import close, load

pytestmark = pytest.mark.gpu


def test_l1_ssim_psnr_match_reference_fixture(hip_device):
    from mobgs_amd.loss_utils import l1_loss, photometric_loss, psnr, ssim
    fx = load("losses")
    gt = torch.from_numpy(fx["gt"]).to(hip_device)
    img = torch.from_numpy(fx["img"]).to(hip_device).requires_grad_(True)
    close(l1_loss(img, gt), fx["l1"], 1e-6, 1e-7, "l1")
    close(ssim(img, gt), fx["ssim"], 1e-5, 1e-6, "ssim")
    close(ssim(img, gt, size_average=False), fx["ssim_per_image"], 1e-5, 1e-6, "ssim per image")
    close(psnr(img.detach(), gt), fx["psnr"], 1e-6, 1e-5, "psnr")
    loss = photometric_loss(img, gt, 0.2)
    close(loss, fx["loss"], 1e-6, 1e-6, "photo loss")
    loss.backward()
    ref = fx["grad_img"]
    close(img.grad, ref, 1e-4, 1e-5 * float(np.abs(ref).max()), "d loss / d image")
    # separate calls give the same gradient as the fused one
    img2 = torch.from_numpy(fx["img"]).to(hip_device).requires_grad_(True)
    (l1_loss(img2, gt) + 0.2 * (1.0 - ssim(img2, gt))).backward()
    close(img2.grad, ref, 1e-4, 1e-5 * float(np.abs(ref).max()), "d loss / d image (separate calls)")


def test_fullsize_loss_speed_smoke(hip_device):
    from mobgs_amd.loss_utils import photometric_loss
    g = torch.Generator().manual_seed(0)
    gt = torch.rand(1, 3, 1014, 1352, generator=g).to(hip_device)
    img = torch.rand(1, 3, 1014, 1352, generator=g).to(hip_device).requires_grad_(True)
    loss = photometric_loss(img, gt)
    loss.backward()
    assert torch.isfinite(loss) and torch.isfinite(img.grad).all()
    assert 0.2 < float(loss) < 0.6  # two independent uniform images: L1 = 1/3, SSIM ~ 0


def test_get_normals_matches_reference_fixture(hip_device):
    """mobgs_amd.main_utils.get_normals (csrc/normals.hip) against the reference's own function: values and the
    gradient w.r.t. the depth map (tests/golden/normals.npz)."""
    import types
    from mobgs_amd.main_utils import get_normals
    fx = load("normals")
    fxv, fyv, cx, cy, skew = (float(v) for v in fx["intrinsics"])
    meta = types.SimpleNamespace(scale_factor_x=fxv, scale_factor_y=fyv, principal_point_x=cx,
                                 principal_point_y=cy, skew=skew, use_center=True)
    z = torch.from_numpy(fx["z"]).to(hip_device).requires_grad_(True)
    n = get_normals(z + 1e-6, meta)
    ref = torch.from_numpy(fx["normals"])
    assert n.shape == ref.shape
    assert torch.allclose(n.detach().cpu(), ref, rtol=0, atol=5e-6), float((n.detach().cpu() - ref).abs().max())
    (n * torch.from_numpy(fx["cotangent"]).to(hip_device)).sum().backward()
    gref = torch.from_numpy(fx["grad_z"])
    err = (z.grad.cpu() - gref).abs().max()
    assert torch.allclose(z.grad.cpu(), gref, rtol=1e-3, atol=2e-5 * float(gref.abs().max())), float(err)
    # border is exactly zero, interior unit length
    assert float(n[0, :, 0, :].abs().max()) == 0.0 and float(n[0, :, :, -1].abs().max()) == 0.0
    assert torch.allclose(n[0, :, 1:-1, 1:-1].norm(dim=0), torch.ones(1, device=hip_device), atol=1e-5)
