"""The host fast path (csrc/fastpath.cpp: the autograd-node bodies in C++) against the Python bodies it mirrors: both
end in the same C-ABI calls, so every output and gradient must be bit-identical."""
import numpy as np
import pytest
import torch

from helpers import leaf_map, load, scene_from_fixture

pytestmark = pytest.mark.gpu


@pytest.fixture()
def fast_switch():
    from mobgs_amd import _fast
    yield _fast
    _fast.reset(None)


def _run(fx, dev, sink, train_mode, half=False):
    import mobgs_amd.gaussian_renderer as GR
    from mobgs_amd.ops import LeafGradSink
    import contextlib
    cam, stat, dyn, bg, _ = scene_from_fixture(fx, device=dev)
    if half:
        for pc in (stat, dyn):
            for n in ("_scaling", "_rotation", "_opacity", "_features_dc", "_features_t", "_omega"):
                if hasattr(pc, n) and getattr(pc, n) is not None:
                    setattr(pc, n, getattr(pc, n).detach().half().requires_grad_(True))
    g = torch.Generator().manual_seed(5)
    H, W = int(cam.image_height), int(cam.image_width)
    v3, v1 = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
    outs = []
    for d in (None, 0.25, -0.3):
        o = GR.render(cam, stat, dyn, None, bg, delta_exposure=d, get_static=train_mode, get_dynamic=train_mode)
        outs.append(o)
    loss = sum((o["render"] * v3).sum() * (i + 1) + (o["depth"] * v1).sum() for i, o in enumerate(outs))
    if train_mode:
        loss = loss + (outs[0]["d_render"] * v3).sum() + (outs[0]["s_alpha"] * v1).sum()
    with (LeafGradSink(stat, dyn) if sink else contextlib.nullcontext()):
        loss.backward()
    res = {"render%d" % i: o["render"].detach().cpu() for i, o in enumerate(outs)}
    res.update({"depth%d" % i: o["depth"].detach().cpu() for i, o in enumerate(outs)})
    res["radii"] = outs[0]["radii"].cpu()
    res["vsp"] = outs[1]["viewspace_points"].grad.cpu()
    for k, t in leaf_map(stat, dyn).items():
        if t.grad is not None:
            res["grad_" + k] = t.grad.detach().float().cpu()
    return res


@pytest.mark.parametrize("sink,train_mode,half", [(False, False, False), (True, False, False), (True, True, False),
                                                  (False, False, True)])
def test_fast_bodies_equal_python_bodies(hip_device, fast_switch, sink, train_mode, half):
    fx = load("render_train")
    fast_switch.reset(True)
    assert fast_switch.get() is not None, f"host fast path not loaded: {fast_switch.load_error}"
    a = _run(fx, hip_device, sink, train_mode, half)
    fast_switch.reset(False)
    assert fast_switch.get() is None
    b = _run(fx, hip_device, sink, train_mode, half)
    assert set(a) == set(b)
    for k in a:
        assert torch.equal(a[k], b[k]), k


def test_fast_path_is_what_runs_on_the_gpu_box(hip_device):
    """The default configuration on a HIP device uses the C++ bodies (no silent fallback to the Python ones)."""
    from mobgs_amd import _fast
    assert _fast.enabled and _fast.get() is not None, _fast.load_error


def test_fast_path_reports_c_abi_errors(hip_device, fast_switch):
    fast_switch.reset(True)
    F = fast_switch.get()
    with pytest.raises(RuntimeError, match="HIP device"):
        F.project_bwd(16, 16, 0.3, torch.zeros(4, 3), torch.zeros(4, 4), torch.zeros(4, 3), torch.zeros(1, 4, 4),
                      torch.zeros(1, 3, 3), torch.zeros(1, 4, dtype=torch.int32), torch.zeros(1, 4, 3), None, None,
                      None, 0)
