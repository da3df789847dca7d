"""BLCE forward API against the fixture produced by the reference's own scene/blce.py (runs on CPU: this part of
the path is plain PyTorch by design, see mobgs_amd/blce.py)."""
import numpy as np
import torch

from helpers import close, load
from mobgs_amd.blce import BLCE, blceKernel, compute_frequency_blur_feature
from mobgs_amd.camera import PinholeCamera


def _model(fx):
    m = BLCE(num_views=3, view_dim=32, num_warp=9)
    sd = {k[3:]: torch.from_numpy(v) for k, v in fx.items() if k.startswith("sd_")}
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m


def test_state_dict_layout_and_forward_match_reference_fixture():
    fx = load("blce")
    m = _model(fx)  # strict load: every reference key exists with the same shape
    assert sum(p.numel() for p in m.parameters()) == int(fx["n_params"][0])
    blur = compute_frequency_blur_feature(torch.from_numpy(fx["in_image"]))
    close(blur, fx["out_blur"], 1e-6, 1e-7, "blur feature")
    Rt_new, expo = m(torch.from_numpy(fx["in_c2w"]), blur, int(fx["in_idx"][0]))
    close(Rt_new, fx["out_Rt_new"], 1e-5, 1e-6, "Rt_new")
    close(expo, fx["out_exposure"], 1e-6, 1e-7, "exposure_time")
    (Rt_new * torch.from_numpy(fx["cot"])).sum().backward()
    n = 0
    for k, p in m.named_parameters():
        if "grad_" + k in fx:
            ref = fx["grad_" + k]
            close(p.grad, ref, 1e-4, 1e-5 * float(np.abs(ref).max()) + 1e-9, f"grad {k}")
            n += 1
    assert n >= 10


def test_get_warped_cams_contract():
    W, H = 40, 30
    K = torch.tensor([[35.0, 0, 20], [0, 35.0, 15], [0, 0, 1]])
    w2c = torch.eye(4)
    w2c[:3, 3] = torch.tensor([0.1, -0.2, 0.3])
    cam = PinholeCamera(W, H, K, w2c, time=0.5, max_time=23)
    cam.uid = 0
    cam.image = torch.rand(3, H, W, generator=torch.Generator().manual_seed(0))
    kern = blceKernel(num_views=2, num_warp=9, iteration=10000)
    cams, expo = kern.get_warped_cams(cam, cam, cam)
    assert len(cams) == 9 and expo.shape == (9,)
    assert torch.allclose(expo, torch.linspace(-0.4, 0.4, 9))
    # near-identity initialisation (decoder gain 1e-5): the warped poses start at the view's own pose
    for c in cams:
        assert torch.allclose(c.world_view_transform.transpose(0, 1), w2c, atol=1e-4)
        assert c.time == cam.time and c.max_time == cam.max_time and c.image_width == W and c.uid == 0
    ray = cams[4].cam_ray
    assert ray.shape == (1, 6, H, W) and ray.requires_grad  # pose gradients reach the BLCE parameters
    assert torch.allclose(ray.detach(), cam.cam_ray, atol=1e-3)
    ray.sum().backward()
    assert kern.model.rot_decoder[0].weight.grad is not None
    kern.optimizer.step()
    kern.adjust_lr()


def test_fused_kernel_is_gated_on_the_shapes_it_hard_codes():
    """csrc/blce.hip reads the 22 parameter tensors of a view with the strides of the reference's default sizes;
    `blceopt.view_dim` is a config option (train.py:260).  Any other shape must take the PyTorch path."""
    from mobgs_amd.blce import _fused_shapes_ok
    assert _fused_shapes_ok(BLCE(num_views=3, view_dim=32, num_warp=9))
    assert not _fused_shapes_ok(BLCE(num_views=3, view_dim=64, num_warp=9))
    assert not _fused_shapes_ok(BLCE(num_views=3, view_dim=32, num_warp=5))
