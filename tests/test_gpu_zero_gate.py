"""Zero-cotangent gate of the backward compositing passes (MobgsTuning.gate_zero_cotangent, round 5).

get_flow()'s images feed one loss term whose weight is 0 in the shipped configurations
(/root/reference/train.py:675, arguments/stereo/seesaw.py:18): the calls are made, the cotangents arrive as exact zeros.
The nodes get_flow() records probe their cotangents on the device and skip the pass when all of them are zero.
Checked here: gradients BIT-IDENTICAL to the ungated path whenever some cotangent is non-zero (also when only the alpha
output carries one, and when the only non-zero element is the very last one), EXACT zeros when all are zero, and that the
skipped pass really costs less than the pass."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _scene(dev, W, H, ns, nd):
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.helper_model import Sandwich
    from mobgs_amd.synth import SynthCamera, dynamic_extras, gaussian_cloud
    scam = SynthCamera().scaled(W, H)
    stat_p, dyn_p = gaussian_cloud(ns, scam, 0), gaussian_cloud(nd, scam, 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], 0)
    torch.manual_seed(0)
    dec = Sandwich(9, 3).to(dev)
    stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
    dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
    pose = torch.eye(4)
    pose[0, 3], pose[2, 3] = 0.05, 0.1
    cam = PinholeCamera(W, H, scam.K, pose, scam.time, scam.max_time, device=dev)
    return cam, stat, dyn


def _operator_inputs(dev, W, H, n, channels, seed=0):
    from mobgs_amd.synth import SynthCamera, gaussian_cloud
    import mobgs_amd.rendering as R
    scam = SynthCamera().scaled(W, H)
    p = gaussian_cloud(n, scam, seed)
    g = torch.Generator().manual_seed(seed)
    means = p["xyz"].to(dev)
    quats = p["rotation"].to(dev)
    scales = torch.exp(p["scaling"]).to(dev)
    opac = torch.sigmoid(p["opacity"]).reshape(-1).to(dev)
    colors = torch.rand(n, channels, generator=g).to(dev)
    viewmat = torch.eye(4, device=dev)[None]
    K = scam.K.to(dev)[None]
    sp = R.SharedProjection(means, quats, scales, opac, viewmat, K, W, H)
    return sp, colors, opac


def _backward(sp, colors, opac, W, H, v_img, v_alpha, gate):
    import mobgs_amd.rendering as R
    m2d = sp.means2d.detach().requires_grad_(True)
    con = sp.conics.detach().requires_grad_(True)
    col = colors.clone().requires_grad_(True)
    op = opac.clone().requires_grad_(True)
    with R.zero_cotangent_gate(gate):
        img, alpha = R.rasterize_to_pixels(m2d, con, col, op, sp.radii, sp.tl, W, H)
    loss = 0.0
    if v_img is not None:
        loss = loss + (img * v_img).sum()
    if v_alpha is not None:
        loss = loss + (alpha * v_alpha).sum()
    loss.backward()
    return [m2d.grad, con.grad, col.grad, op.grad]


@pytest.mark.parametrize("channels", [3, 10, 12])
def test_gated_pass_is_bit_identical_or_exactly_zero(hip_device, channels):
    dev = hip_device
    W, H, n = 500, 300, 20_000
    sp, colors, opac = _operator_inputs(dev, W, H, n, channels)
    g = torch.Generator().manual_seed(3)
    v_img = torch.randn(1, H, W, channels, generator=g).to(dev)
    v_alpha = torch.randn(1, H, W, 1, generator=g).to(dev)
    last_only = torch.zeros_like(v_img)
    last_only[0, H - 1, W - 1, channels - 1] = 1.0   # the probe's tail handling: one non-zero element at the very end
    first_only = torch.zeros_like(v_img)
    first_only[0, 0, 0, 0] = -0.5
    for vi, va in ((v_img, v_alpha), (v_img, None), (None, v_alpha), (torch.zeros_like(v_img), v_alpha),
                   (last_only, None), (first_only, torch.zeros_like(v_alpha))):
        ref = _backward(sp, colors, opac, W, H, vi, va, gate=False)
        got = _backward(sp, colors, opac, W, H, vi, va, gate=True)
        assert any(float(r.abs().max()) > 0 for r in ref)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    # all cotangents exactly zero (a zero loss weight): exact zeros, whether -0.0 or +0.0 arrives
    for z in (0.0, -0.0):
        got = _backward(sp, colors, opac, W, H, torch.full_like(v_img, z), torch.full_like(v_alpha, z), gate=True)
        for t in got:
            assert t is not None and float(t.abs().max()) == 0.0
    # a NaN is not a zero: the pass runs and says so
    nan_img = torch.zeros_like(v_img)
    nan_img[0, H // 2, W // 2, 0] = float("nan")
    got = _backward(sp, colors, opac, W, H, nan_img, None, gate=True)
    assert any(bool(torch.isnan(t).any()) for t in got)


def test_skipped_pass_costs_no_more_than_the_pass(hip_device):
    dev = hip_device
    W, H, n = 1352, 1014, 200_000
    sp, colors, opac = _operator_inputs(dev, W, H, n, 10)
    zero_img = torch.zeros(1, H, W, 10, device=dev)
    ms = {}
    for gate in (False, True):
        for _ in range(2):
            _backward(sp, colors, opac, W, H, zero_img, None, gate)
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(5):
            _backward(sp, colors, opac, W, H, zero_img, None, gate)
        t1.record()
        torch.cuda.synchronize()
        ms[gate] = t0.elapsed_time(t1) / 5
    print(f"\n[zero gate] forward + backward with zero cotangents: ungated {ms[False]:.3f} ms, gated {ms[True]:.3f} ms")
    # (the ungated pass is cheap too when every cotangent is zero -- no pixel has a live contributor -- but it still
    # walks the tiles; the gate adds one probe launch and must not cost noticeably more than it saves)
    assert ms[True] < 1.15 * ms[False]


def _flow_run(dev, W, H, deltas, ws, weight, device_gate, host_gate, separate=False):
    import mobgs_amd.gaussian_renderer as G
    G.ZERO_GATE, G.FLOW_HOST_GATE = device_gate, host_gate
    G.invalidate_flow_cache()
    try:
        cam, stat, dyn = _scene(dev, W, H, 6_000, 3_000)
        bg = torch.zeros(9, device=dev)
        if separate:
            outs = [G.get_flow(cam, stat, dyn, None, bg, delta_exposure=d) for d in deltas]
        else:
            outs = G.get_flow_many(cam, stat, dyn, None, bg, deltas)
        loss = 0.0
        for o, w in zip(outs, ws):
            for t, wt in zip(o, w):
                loss = loss + (t * wt).sum()
        (loss * weight).backward()
        return [stat._xyz.grad, stat._scaling.grad, stat._features_dc.grad, dyn.control_xyz.grad, dyn._rotation.grad,
                dyn._omega.grad, dyn._opacity.grad, dyn._features_t.grad, dyn.rgbdecoder.mlp1.weight.grad]
    finally:
        G.ZERO_GATE, G.FLOW_HOST_GATE = True, True


def test_get_flow_gradients_with_and_without_the_gates(hip_device):
    dev = hip_device
    W, H = 320, 200
    deltas = [-0.4, -0.2, 0.0, 0.3]
    g = torch.Generator().manual_seed(5)
    ws = [[torch.randn(*s, generator=g).to(dev) for s in ((1, H, W, 2), (1, H, W, 2), (3, H, W), (1, H, W))] for _ in deltas]
    ref = _flow_run(dev, W, H, deltas, ws, 1.0, False, False)
    assert float(ref[0].abs().max()) > 0
    for device_gate, host_gate in ((True, False), (True, True)):
        got = _flow_run(dev, W, H, deltas, ws, 1.0, device_gate, host_gate)
        for a, b in zip(ref, got):
            assert torch.equal(a, b)
    # weight 0 (train.py:675 with lambda_flow_loss = 0).  Device gate alone: every leaf gets an exactly-zero gradient;
    # with the host-side head the sub-graph is not entered at all: the term adds nothing (.grad stays None here)
    for t in _flow_run(dev, W, H, deltas, ws, 0.0, True, False):
        assert t is not None and float(t.abs().max()) == 0.0
    for t in _flow_run(dev, W, H, deltas, ws, 0.0, True, True):
        assert t is None or float(t.abs().max()) == 0.0
    # a zero-weight flow term next to a live one: the live gradients are those of the live term alone
    ws0 = [[torch.zeros_like(w) for w in row] for row in ws[:-1]] + [ws[-1]]
    a = _flow_run(dev, W, H, deltas, ws0, 1.0, False, False)
    b = _flow_run(dev, W, H, deltas, ws0, 1.0, True, True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def test_separate_get_flow_calls_share_the_mid_exposure_state(hip_device):
    """The unchanged caller's nine get_flow() calls per view (train.py:570-579): the mid-exposure state is built once
    (implicit cache), results equal the uncached calls' -- images bit for bit, gradients to summation order (the shared
    state's gradient is one sum instead of one per call) -- and the cache notices an optimiser step."""
    import mobgs_amd.gaussian_renderer as G
    from helpers import close
    dev = hip_device
    W, H = 320, 200
    deltas = [-0.5, 0.25, 0.5]
    g = torch.Generator().manual_seed(6)
    ws = [[torch.randn(*s, generator=g).to(dev) for s in ((1, H, W, 2), (1, H, W, 2), (3, H, W), (1, H, W))] for _ in deltas]
    G.FLOW_MID_CACHE = False
    try:
        ref = _flow_run(dev, W, H, deltas, ws, 1.0, True, True, separate=True)
    finally:
        G.FLOW_MID_CACHE = True
    h0, m0 = G.mid_cache_stats["hits"], G.mid_cache_stats["misses"]
    got = _flow_run(dev, W, H, deltas, ws, 1.0, True, True, separate=True)
    assert G.mid_cache_stats["hits"] - h0 == len(deltas) - 1 and G.mid_cache_stats["misses"] - m0 == 1
    for a, b in zip(ref, got):
        close(b, a, 2e-5 * float(a.abs().max()), 1e-4, "shared mid state")
    # forward values are bit-identical, and a parameter update invalidates the entry
    cam, stat, dyn = _scene(dev, W, H, 6_000, 3_000)
    bg = torch.zeros(9, device=dev)
    G.invalidate_flow_cache()
    with torch.no_grad():
        o1 = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
        h1 = G.mid_cache_stats["hits"]
        o2 = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
        assert G.mid_cache_stats["hits"] == h1 + 1
        for x, y in zip(o1, o2):
            assert torch.equal(x, y)
        dyn.control_xyz.add_(0.01)          # what an optimiser step does: in place, version counter bumped
        m1 = G.mid_cache_stats["misses"]
        o3 = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
        assert G.mid_cache_stats["misses"] == m1 + 1
        assert not torch.equal(o3[1], o2[1])
    # a backward pass through the shared state retires it: the next call builds a fresh one instead of a freed graph
    G.invalidate_flow_cache()
    cam, stat, dyn = _scene(dev, W, H, 6_000, 3_000)
    o = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
    (o[1] * ws[0][1]).sum().backward()
    m2 = G.mid_cache_stats["misses"]
    o = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
    assert G.mid_cache_stats["misses"] == m2 + 1
    (o[1] * ws[0][1]).sum().backward()


def test_mid_state_cache_sees_the_fused_adam_step_and_row_permutations(hip_device):
    """optim.fused_adam_step writes the parameters through raw pointers -- Tensor._version does not
    move -- and GaussianParams.spatial_sort_ swaps `.data`; the implicit mid-exposure cache of separate get_flow() calls
    keyed on (storage, version, shape) would hand back the projection of the OLD values.  Both now announce themselves
    (gaussian_renderer.parameters_changed): the next call is a miss and renders the new parameters."""
    import mobgs_amd.gaussian_renderer as G
    from mobgs_amd.optim import fused_adam_step
    dev = hip_device
    W, H = 320, 200
    cam, stat, dyn = _scene(dev, W, H, 6_000, 3_000)
    bg = torch.zeros(9, device=dev)
    opt = torch.optim.Adam([dyn.control_xyz, stat._xyz], lr=0.05)
    G.invalidate_flow_cache()
    with torch.no_grad():
        o1 = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
    ver = (dyn.control_xyz._version, stat._xyz._version)
    dyn.control_xyz.grad = torch.ones_like(dyn.control_xyz)
    stat._xyz.grad = torch.ones_like(stat._xyz)
    assert fused_adam_step([opt]) == 2
    assert (dyn.control_xyz._version, stat._xyz._version) == ver, "(the premise: the fused step does not bump versions)"
    m0 = G.mid_cache_stats["misses"]
    with torch.no_grad():
        o2 = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
        G.FLOW_MID_CACHE = False
        try:
            ref = G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
        finally:
            G.FLOW_MID_CACHE = True
    assert G.mid_cache_stats["misses"] == m0 + 1, "the cache must not survive a fused optimiser step"
    assert not torch.equal(o2[1], o1[1])
    for x, y in zip(o2, ref):
        assert torch.equal(x, y)
    # a row permutation (same values, other rows): a miss again, same images up to the order of equal-depth ties
    with torch.no_grad():
        G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
    m1 = G.mid_cache_stats["misses"]
    stat.spatial_sort_()
    dyn.spatial_sort_()
    with torch.no_grad():
        G.get_flow(cam, stat, dyn, None, bg, delta_exposure=0.25)
    assert G.mid_cache_stats["misses"] == m1 + 1
