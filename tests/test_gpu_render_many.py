"""render_many(): the K latent sub-frames of a blurry view as ONE batch of K cameras with per-camera geometry
(MobgsTuning.geometry_per_camera) against K separate render() calls."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("size,ns,nd", [((232, 120), 900, 500), ((512, 288), 20_000, 10_000),
                                        ((1352, 1014), 200_000, 100_000)])   # ... and BASELINE config #2's size
def test_render_many_equals_separate_renders(hip_device, size, ns, nd):
    import bench as B
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_renderer import render, render_many
    dev = hip_device
    W, H = size
    K = 5 if W < 1000 else 3
    scam, cam0, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H, seed=4)
    cams = [PinholeCamera(W, H, scam.K, B.view_pose(k), time=scam.time, max_time=scam.max_time, device=dev)
            for k in range(K)]
    deltas = [torch.tensor(0.3 * (k - K // 2), device=dev) for k in range(K)]
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
    from mobgs_amd.gaussian_renderer import viewspace_grad
    ref_vs = [viewspace_grad(o).clone() for o in ref]   # what train.py:637-646 reads per render
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
    # the densification input of every sub-frame: the batch shares ONE [K,N,2] tensor, viewspace_grad() picks the row
    for k, (o, gr) in enumerate(zip(outs, ref_vs)):
        assert o["viewspace_points"].shape[0] == K and o["viewspace_index"] == k
        got = viewspace_grad(o)
        assert got.shape == gr.shape
        sc = float(gr.abs().max())
        assert torch.allclose(got, gr, rtol=1e-4, atol=2e-5 * sc + 1e-12), f"viewspace gradient of sub-frame {k}"


def test_prep_of_k_instants_in_one_launch_equals_k_launches(hip_device):
    """ops.PrepSplats with times [K,2] (mobgs_prep_{fwd,bwd}_many): row block k of means / quats / colours is the
    single-instant result bit for bit, and the leaf gradients are those of K single-instant backward passes accumulated
    in instant order -- bit for bit as well (LeafGradSink buffers: instant 0 writes, the others add)."""
    import bench as B
    from mobgs_amd.gaussian_renderer import _prep
    dev = hip_device
    _, _, stat, dyn, _ = B.build_scene(dev, 700, 400, 64, 48, seed=7)
    K = 4
    g = torch.Generator().manual_seed(5)
    times = torch.tensor([[0.21, 0.21], [0.3, 0.3], [1.07, 1.0], [-0.04, 0.0]], device=dev)
    N = 1100
    cot = [torch.randn(K, N, 3, generator=g).to(dev), torch.randn(K, N, 4, generator=g).to(dev),
           torch.randn(N, 3, generator=g).to(dev), torch.randn(N, generator=g).to(dev),
           torch.randn(K, N, 9, generator=g).to(dev)]
    params = B.leaves(stat, dyn)

    def run(batched):
        for p in params:
            p.grad = None
        if batched:
            out = _prep(stat, dyn, times)
            torch.autograd.backward(list(out), cot)
        else:
            outs = [_prep(stat, dyn, times[k]) for k in range(K)]
            heads, grads = [], []
            for k, o in enumerate(outs):   # scales / opacities: the cotangent exists once -> on instant 0
                heads += [o[0], o[1], o[4]] + ([o[2], o[3]] if k == 0 else [])
                grads += [cot[0][k], cot[1][k], cot[4][k]] + ([cot[2], cot[3]] if k == 0 else [])
            torch.autograd.backward(heads, grads)
            out = (torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs]), outs[0][2], outs[0][3],
                   torch.stack([o[4] for o in outs]))
        return [t.detach().clone() for t in out], [p.grad.clone() for p in params if p.grad is not None]

    a, ga = run(True)
    b, gb = run(False)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    assert len(ga) == len(gb) and len(ga) >= 13
    for i, (x, y) in enumerate(zip(ga, gb)):
        # autograd sums the K single-instant gradients pairwise in its own order; the batched kernel adds them in
        # instant order: equal to rounding of K - 1 additions
        assert torch.allclose(x, y, rtol=1e-5, atol=1e-6 * float(y.abs().max()) + 1e-12), (i, float((x - y).abs().max()))


def test_batched_decode_equals_image_by_image(hip_device):
    """ops.decode / decode_with_channels on a batch [C,H,W,CF] (mobgs_decoder_{fwd,bwd}_many): images bit-identical to C
    single-image calls; per-image pose gradients equal; weight gradients equal to the order of the sum over images; a
    shared pose that needs a gradient is expanded per image."""
    from mobgs_amd.ops import decode, decode_with_channels
    dev = hip_device
    g = torch.Generator().manual_seed(11)
    C, H, W = 3, 37, 53
    feat = torch.randn(C, H, W, 12, generator=g).to(dev).requires_grad_(True)
    alphas = (0.2 + 0.8 * torch.rand(C, H, W, 1, generator=g)).to(dev).requires_grad_(True)
    w1 = (0.3 * torch.randn(6, 12, generator=g)).to(dev).requires_grad_(True)
    w2 = (0.3 * torch.randn(3, 6, generator=g)).to(dev).requires_grad_(True)
    intr = torch.tensor([60.0, 55.0, 26.0, 18.0], device=dev)
    poses = []
    for c in range(C):
        m = torch.eye(4)
        m[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        m[:3, 3] = torch.randn(3, generator=g)
        poses.append(m.to(dev).requires_grad_(True))
    v_rgb = torch.randn(C, 3, H, W, generator=g).to(dev)
    v_dep = torch.randn(C, H, W, generator=g).to(dev)
    leaves = [feat, alphas, w1, w2] + poses

    def grads():
        out = [t.grad.clone() if t.grad is not None else None for t in leaves]
        for t in leaves:
            t.grad = None
        return out

    single = [decode(feat[c], alphas[c], (intr, poses[c]), w1, w2, True) for c in range(C)]
    torch.autograd.backward([s[0] for s in single] + [s[1] for s in single], list(v_rgb) + list(v_dep))
    g_single = grads()
    rgb, dep = decode(feat, alphas, (intr.expand(C, 4).contiguous(), torch.stack(poses)), w1, w2, True)
    assert rgb.shape == (C, 3, H, W) and dep.shape == (C, H, W)
    for c in range(C):
        assert torch.equal(rgb[c], single[c][0]) and torch.equal(dep[c], single[c][1])
    torch.autograd.backward([rgb, dep], [v_rgb, v_dep])
    g_batch = grads()
    for i, (a, b) in enumerate(zip(g_batch, g_single)):
        if i in (2, 3):   # weights: one reduction over all images instead of C accumulated ones
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max())), i
        else:
            assert torch.equal(a, b), i
    # shared pose with a gradient + the channel hand-out of get_flow
    shared = poses[0]
    r1, ch1 = decode_with_channels(feat, None, (intr, shared), w1, w2, 9, 2)
    assert r1.shape == (C, 3, H, W) and ch1.shape == (C, H, W, 2)
    torch.autograd.backward([r1, ch1], [v_rgb, torch.ones_like(ch1)])
    g_sh = shared.grad.clone()
    feat_grad = feat.grad.clone()
    grads()
    outs = [decode_with_channels(feat[c], None, (intr, shared), w1, w2, 9, 2) for c in range(C)]
    torch.autograd.backward([o[0] for o in outs] + [o[1] for o in outs], list(v_rgb) + [torch.ones(H, W, 2, device=dev)] * C)
    assert torch.allclose(g_sh, shared.grad, rtol=1e-5, atol=1e-6 * float(shared.grad.abs().max()))
    assert torch.equal(feat_grad, feat.grad)
    for c in range(C):
        assert torch.equal(r1[c], outs[c][0])


def test_render_many_of_one_camera_is_render(hip_device):
    """K = 1 (a rank that owns a single latent sub-frame of a view in a sharded run): the batch of one goes through the
    single-image calls and equals render()."""
    import bench as B
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_renderer import render, render_many
    from mobgs_amd.ops import decode
    dev = hip_device
    W, H = 96, 64
    scam, _, stat, dyn, _ = B.build_scene(dev, 500, 300, W, H, seed=9)
    cam = PinholeCamera(W, H, scam.K, B.view_pose(1), time=scam.time, max_time=scam.max_time, device=dev)
    bg = torch.zeros(9, device=dev)
    d = torch.tensor(0.2, device=dev)
    ref = render(cam, stat, dyn, None, bg, delta_exposure=d)
    out, = render_many([cam], stat, dyn, None, bg, [d])
    assert out["render"].shape == (3, H, W) and torch.equal(out["render"], ref["render"])
    assert torch.equal(out["depth"], ref["depth"])
    (out["render"].sum() + out["depth"].sum()).backward()
    # stacked parameters of one camera are accepted by the decoder wrapper as that camera's
    feat = torch.randn(1, H, W, 10, device=dev)
    a = torch.rand(1, H, W, 1, device=dev) + 0.1
    w1, w2 = torch.randn(6, 12, device=dev), torch.randn(3, 6, device=dev)
    r1, d1 = decode(feat, a, (cam.ray_intrinsics[None], cam.ray_c2w[None]), w1, w2, True)
    r0, d0 = decode(feat, a, (cam.ray_intrinsics, cam.ray_c2w), w1, w2, True)
    assert r1.shape == (3, H, W) and torch.equal(r1, r0) and torch.equal(d1, d0)
