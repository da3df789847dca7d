"""render_many(): the K latent sub-frames of a blurry view as ONE batch of K cameras with per-camera geometry
(MobgsTuning.geometry_per_camera) against K separate render() calls."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,ns,nd", [((232, 120), 900, 500), ((512, 288), 20_000, 10_000)])
def test_render_many_equals_separate_renders(hip_device, size, ns, nd):
    import bench as B
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_renderer import render, render_many
    dev = hip_device
    W, H = size
    K = 5
    scam, cam0, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H, seed=4)
    cams = [PinholeCamera(W, H, scam.K, B.view_pose(k), time=scam.time, max_time=scam.max_time, device=dev)
            for k in range(K)]
    deltas = [torch.tensor(0.3 * (k - 2), device=dev) for k in range(K)]
    bg = torch.zeros(9, device=dev)
    g = torch.Generator().manual_seed(1)
    v3 = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(K)]
    v1 = [torch.randn(1, H, W, generator=g).to(dev) for _ in range(K)]
    params = B.leaves(stat, dyn)

    def grads():
        return [p.grad.clone() for p in params]

    for p in params:
        p.grad = None
    ref = [render(c, stat, dyn, None, bg, delta_exposure=d) for c, d in zip(cams, deltas)]
    torch.autograd.backward([o["render"] for o in ref] + [o["depth"] for o in ref], v3 + v1)
    ref_out = [(o["render"].detach().clone(), o["depth"].detach().clone(), o["radii"].clone()) for o in ref]
    ref_g = grads()
    del ref
    for p in params:
        p.grad = None
    outs = render_many(cams, stat, dyn, None, bg, deltas)
    torch.autograd.backward([o["render"] for o in outs] + [o["depth"] for o in outs], v3 + v1)
    for k, (o, (r, d, rad)) in enumerate(zip(outs, ref_out)):
        assert torch.equal(o["radii"], rad), f"radii of sub-frame {k}"
        assert torch.equal(o["render"], r), f"image of sub-frame {k}"
        assert torch.equal(o["depth"], d), f"depth of sub-frame {k}"
    for i, (p, gr) in enumerate(zip(params, ref_g)):
        sc = float(gr.abs().max())
        assert torch.allclose(p.grad, gr, rtol=1e-4, atol=2e-5 * sc + 1e-12), \
            f"leaf {i}: max err {float((p.grad - gr).abs().max()):.3e} of {sc:.3e}"
