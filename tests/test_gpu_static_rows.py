"""MobgsTuning.static_rows (round 6): the reference's STATIC splats carry colour features
cat(f_dc, 0.0 * f_t) (/root/reference/scene/gaussian_model.py:244-246) -- three structurally-zero channels in render()'s
10-channel pass, five in get_flow()'s 12-channel pass (their flow is x - x = 0, gaussian_renderer/__init__.py:436-476).
The backward compositor raster_bwd_kernel takes a blend body without those channels for such entries (one wave-uniform
branch per list entry).  What must hold:
  * render(): every output and every leaf gradient BIT-identical with the statement on and off (fma(0, v, acc) = acc; the
    dead channels' own gradient is multiplied by 0.0 in the prep backward) -- on the benchmark's kernel selection (grids
    of > 1024 tiles, or forced) and in train mode (class passes);
  * the statement is about rows, the zeros are VERIFIED: a "static" row whose channels 6..8 are not zero gets the full
    body and its full gradient;
  * get_flow(): outputs identical; gradients equal up to the rounding of terms that cancel (+g and -g of a static
    splat's flow through two identical projections)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run_render(dev, W, H, ns, nd, on, train=False, headline=False):
    import bench as B
    from mobgs_amd import rendering
    from mobgs_amd.gaussian_renderer import render
    saved = (rendering.STATIC_ROWS, rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma, rendering.path_log)
    rendering.STATIC_ROWS = on
    if headline:
        rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma = 0, 0
    rendering.path_log = []
    try:
        scam, cam, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H, seed=3)
        cam.world_view_transform.requires_grad_(True)
        g = torch.Generator().manual_seed(9)
        v3, v1 = torch.randn(3, H, W, generator=g).to(dev), torch.randn(1, H, W, generator=g).to(dev)
        outs = None
        for rep in range(2):   # frame 1: single-pass lists (the steady state)
            for p in B.leaves(stat, dyn) + [cam.world_view_transform]:
                p.grad = None
            out = render(cam, stat, dyn, None, torch.zeros(9, device=dev), get_static=train, get_dynamic=train)
            keys = ["render", "depth"] + (["s_render", "d_render", "d_alpha", "s_alpha", "d_depth"] if train else [])
            cots = [v3, v1] + ([v3, v3, v1, v1, v1] if train else [])
            torch.autograd.backward([out[k] for k in keys], cots)
            outs = [out[k].detach().clone() for k in keys] + [out["radii"].clone(), out["viewspace_points"].grad.clone(),
                                                               cam.world_view_transform.grad.clone()]
        grads = [p.grad.clone() for p in B.leaves(stat, dyn)]
        log = [e for e in rendering.path_log if e["dir"] == "bwd" and e["D"] == 10]
        return outs, grads, log, stat.get_xyz.shape[0]
    finally:
        rendering.STATIC_ROWS, rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma, rendering.path_log = saved


@pytest.mark.parametrize("W,H,ns,nd,train,headline", [(704, 400, 30_000, 15_000, False, False),
                                                      (704, 400, 30_000, 15_000, True, False),
                                                      (1352, 1014, 200_000, 100_000, False, False),
                                                      (232, 120, 3_000, 1_500, False, True),
                                                      (232, 120, 3_000, 1_500, True, True)])
def test_render_is_bit_identical_with_the_static_row_body(hip_device, W, H, ns, nd, train, headline):
    res = {on: _run_render(hip_device, W, H, ns, nd, on, train, headline) for on in (False, True)}
    log_on, log_off = res[True][2], res[False][2]
    assert log_on and all(e["bwd_kernel"] == "quadrant" for e in log_on + log_off), log_on
    assert any(e["static_rows"] == ns for e in log_on) and all(e["static_rows"] == 0 for e in log_off)
    for i, (a, b) in enumerate(zip(res[False][0], res[True][0])):
        assert torch.equal(a, b), f"output {i}"
    assert float(res[True][1][0].abs().max()) > 0
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        assert torch.equal(a, b), f"leaf gradient {i}"


def test_zeros_are_verified_per_entry_and_dead_gradients_are_zero(hip_device):
    """Operator level (SharedProjection.composite, 9 colours + depth): rows [0, S) declared static.  Rows whose channels
    6..8 ARE zero: v_colors[:, 6:9] comes back as exact zeros (the full pass returns the non-zero true gradient -- which
    is why only a caller that multiplies it by 0.0 may make the statement); a declared row whose channels are NOT zero
    takes the full body: all of its gradients are bit-identical to the undeclared pass.  Everything else: bit-identical."""
    from mobgs_amd import rendering
    from mobgs_amd.rendering import SharedProjection
    from mobgs_amd.synth import SynthCamera, splat_inputs
    dev = hip_device
    W, H, N, S = 704, 400, 40_000, 25_000
    scam = SynthCamera().scaled(W, H)
    s = splat_inputs(N, scam, 4, 9)
    cols = s["colors"].clone()
    cols[:S, 6:9] = 0.0
    liars = torch.arange(0, S, 7)          # declared static, but with live channels 6..8
    cols[liars, 6:9] = s["colors"][liars, 6:9]
    cols[liars[::2], 7] = 0.0              # ... some of them in one channel only
    v = torch.randn(1, H, W, 10, generator=torch.Generator().manual_seed(5)).to(dev)
    res = {}
    for rows in (0, S):
        t = {k: x.to(dev).clone().requires_grad_(k in ("means", "quats", "scales", "opacities")) for k, x in s.items()}
        c = cols.to(dev).clone().requires_grad_(True)
        sp = SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], t["viewmats"], t["Ks"], W, H)
        sp.static_rows = rows
        img, alphas = sp.composite(c, torch.zeros(1, 9, device=dev))
        (img * v).sum().backward()
        res[rows] = (img.detach().clone(), c.grad.clone(), [t[k].grad.clone() for k in ("means", "quats", "scales",
                                                                                         "opacities")])
    assert torch.equal(res[0][0], res[S][0])
    for a, b in zip(res[0][2], res[S][2]):
        assert torch.equal(a, b)
    full, short = res[0][1], res[S][1]
    honest = torch.ones(N, dtype=torch.bool, device=dev)
    honest[S:] = False
    honest[liars.to(dev)] = False
    assert torch.equal(full[:, :6], short[:, :6])
    assert torch.equal(full[~honest], short[~honest]), "undeclared rows and rows with live channels: the full body"
    assert float(short[honest][:, 6:9].abs().max()) == 0.0
    assert float(full[honest][:, 6:9].abs().max()) > 0.0, "(the true gradient of a zero-valued channel is not zero)"
    assert rendering.STATIC_ROWS


def test_get_flow_with_the_static_row_body(hip_device):
    """12-channel pass (9 features + 2 flow + depth): five dead channels per static entry.  Outputs are bit-identical; a
    static splat's flow gradient +g / -g reaches its parameters through two identical projections and cancels -- with the
    short body it is never formed, so the static leaves' gradients differ by the ROUNDING of that cancellation only."""
    import bench as B
    from mobgs_amd import rendering
    from mobgs_amd.gaussian_renderer import get_flow_many
    dev = hip_device
    W, H, ns, nd = 704, 400, 20_000, 10_000
    res = {}
    for on in (False, True):
        rendering.STATIC_ROWS = on
        rendering.path_log = []
        try:
            scam, cam, stat, dyn, _ = B.build_scene(dev, ns, nd, W, H, seed=6)
            outs = get_flow_many(cam, stat, dyn, None, torch.zeros(9, device=dev), [-0.5, 0.0, 0.25])
            g = torch.Generator().manual_seed(2)
            loss = sum((t * torch.randn(t.shape, generator=g).to(dev)).sum() for o in outs for t in o)
            loss.backward()
            res[on] = ([t.detach().clone() for o in outs for t in o], [p.grad.clone() for p in B.leaves(stat, dyn)],
                       [e for e in rendering.path_log if e["dir"] == "bwd" and e["D"] == 12])
        finally:
            rendering.STATIC_ROWS = True
            rendering.path_log = None
    assert res[True][2] and res[True][2][-1]["static_rows"] == ns and res[False][2][-1]["static_rows"] == 0
    for a, b in zip(res[False][0], res[True][0]):
        assert torch.equal(a, b)
    for i, (a, b) in enumerate(zip(res[False][1], res[True][1])):
        sc = float(a.abs().max())
        assert float((a - b).abs().max()) <= 2e-6 * sc + 1e-12, (i, float((a - b).abs().max()), sc)
