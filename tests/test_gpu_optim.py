"""mobgs_amd.optim.fused_adam_step against torch.optim.Adam (the reference's optimiser, train.py:790-807)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_fused_adam_matches_torch_adam(hip_device):
    from mobgs_amd.optim import fused_adam_step
    dev = hip_device
    g = torch.Generator().manual_seed(3)
    shapes = [(1000, 3), (1000, 4), (1000, 1), (777, 12, 3), (6, 12), (5,), (1,), (333, 6)]
    lrs = [1.6e-4, 1e-3, 5e-2, 5.6e-4, 1e-4, 2.5e-3, 1e-3, 3e-2]

    def make():
        ps = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
        return ps

    torch.manual_seed(0)
    pa = make()
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    # two optimisers with one-tensor groups, as the reference builds them; eps = 1e-15 (gaussian_model.py:645)
    oa = [torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs[:len(ps)]))],
                           lr=0.0, eps=1e-15) for ps in (pa[:5], pa[5:])]
    ob = [torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(ps, lrs[:len(ps)]))],
                           lr=0.0, eps=1e-15) for ps in (pb[:5], pb[5:])]
    for it in range(7):
        grads = [torch.randn(*s, generator=g).to(dev) * (10.0 ** (it % 3 - 1)) for s in shapes]
        for p, q, gr in zip(pa, pb, grads):
            p.grad, q.grad = gr.clone(), gr.clone()
        if it == 3:  # a parameter without a gradient is left alone by both
            pa[2].grad = pb[2].grad = None
        for o in oa:
            o.step()
        assert fused_adam_step(ob) == (len(shapes) - (1 if it == 3 else 0))
        for i, (p, q) in enumerate(zip(pa, pb)):
            # one ulp of the parameter (fp32 contraction of a + alpha * (b / c) may differ between the two kernels)
            assert torch.allclose(q.detach(), p.detach(), rtol=2e-6, atol=2e-7), (it, i, float((p - q).abs().max()))
    for o1, o2 in zip(oa, ob):
        for g1, g2 in zip(o1.param_groups, o2.param_groups):
            s1, s2 = o1.state[g1["params"][0]], o2.state[g2["params"][0]]
            assert float(s1["step"]) == float(s2["step"])
            assert torch.allclose(s2["exp_avg"], s1["exp_avg"], rtol=2e-6, atol=2e-7 * float(s1["exp_avg"].abs().max()))
            assert torch.allclose(s2["exp_avg_sq"], s1["exp_avg_sq"], rtol=2e-6,
                                  atol=2e-7 * float(s1["exp_avg_sq"].abs().max()))


def test_fused_adam_falls_back_for_what_it_does_not_cover(hip_device):
    from mobgs_amd.optim import fused_adam_step
    dev = hip_device
    p32 = torch.randn(50, device=dev, requires_grad=True)
    p16 = torch.randn(50, device=dev).half().requires_grad_(True)
    ref32, ref16 = p32.detach().clone().requires_grad_(True), p16.detach().clone().requires_grad_(True)
    opt = torch.optim.Adam([{"params": [p32], "lr": 1e-2}, {"params": [p16], "lr": 1e-2}], eps=1e-8)
    ref = torch.optim.Adam([{"params": [ref32], "lr": 1e-2}, {"params": [ref16], "lr": 1e-2}], eps=1e-8)
    for _ in range(3):
        g32, g16 = torch.randn(50, device=dev), torch.randn(50, device=dev).half()
        p32.grad, ref32.grad, p16.grad, ref16.grad = g32.clone(), g32.clone(), g16.clone(), g16.clone()
        assert fused_adam_step([opt]) == 1      # the fp32 tensor; the half one goes through torch's own step
        ref.step()
    assert torch.allclose(p32.detach(), ref32.detach(), rtol=2e-6, atol=2e-7)
    assert torch.equal(p16, ref16)


def test_fused_adam_optimizer_class_is_a_drop_in_for_torch_adam(hip_device):
    """optim.FusedAdam -- what TrainableGaussians.training_setup() / blceKernel hand to an unchanged train.py loop
    (`optimizer.step()`, /root/reference/train.py:790-807): torch.optim.Adam's update through ONE launch, torch's own step
    for what the kernel does not cover (a half tensor here), state_dict round trip, zero_grad / param-group edits as usual."""
    from mobgs_amd.optim import FusedAdam
    dev = hip_device
    g = torch.Generator().manual_seed(5)
    shapes = [(500, 3), (500, 12, 3), (6, 12), (40,)]
    pa = [torch.randn(*s, generator=g).to(dev).requires_grad_(True) for s in shapes]
    pa[3] = pa[3].detach().half().requires_grad_(True)
    pb = [p.detach().clone().requires_grad_(True) for p in pa]
    groups = lambda ps: [{"params": [p], "lr": 1e-3 * (i + 1), "name": str(i)} for i, p in enumerate(ps)]  # noqa: E731
    ref, opt = torch.optim.Adam(groups(pa), lr=0.0, eps=1e-15), FusedAdam(groups(pb), lr=0.0, eps=1e-15)
    assert isinstance(opt, torch.optim.Adam)
    for it in range(5):
        for p, q in zip(pa, pb):
            gr = torch.randn(p.shape, generator=g).to(dev).to(p.dtype)
            p.grad, q.grad = gr.clone(), gr.clone()
        ref.step()
        opt.step()
        if it == 2:   # what densification / schedulers do between steps
            for o in (ref, opt):
                o.param_groups[0]["lr"] *= 0.5
            sd = opt.state_dict()
            opt.load_state_dict(sd)
        for i, (p, q) in enumerate(zip(pa, pb)):
            assert torch.allclose(q.detach().float(), p.detach().float(), rtol=2e-6, atol=2e-7 if i < 3 else 1e-3), (it, i)
    opt.zero_grad(set_to_none=True)
    assert all(q.grad is None for q in pb)
    assert float(opt.state[pb[0]]["step"]) == 5.0 and float(opt.state[pb[3]]["step"]) == 5.0
