"""Single-pass tile lists (mobgs_project_and_bin_fused, round 5) against the two-pass path they replace.

The fused path must be indistinguishable to every consumer: the same packed per-tile lists in the same order
(flatten_ids, isect_ids, tile_offsets), the same box-intersection scan (cum_tiles), the same gradient-slot numbering
(keep_scan) and a valid schedule -- with reach culling on AND off (off = gsplat's own lists, which other tests pin to the
C oracle), through the C++ host path and the Python one, with ragged image sizes, several cameras, and when a tile's
list outgrows its key segment (the call must then hand out EMPTY lists and the host must fall back to the two-pass
rebuild and still return the right answer).
"""
import numpy as np
import pytest
import torch

from mobgs_amd.synth import SynthCamera, splat_inputs

pytestmark = pytest.mark.gpu


def _project(s, dev, W, H, fused, hint=None, C=1, order=None):
    import mobgs_amd.rendering as R
    old = R.FUSED_LISTS
    R.FUSED_LISTS = fused
    try:
        t = {k: v.to(dev) for k, v in s.items()}
        vm = t["viewmats"].expand(C, 4, 4).contiguous()
        if C > 1:  # different cameras: shift each one a little
            vm = vm.clone()
            vm[:, 0, 3] += torch.linspace(-0.2, 0.2, C, device=dev)
        Ks = t["Ks"].expand(C, 3, 3).contiguous()
        if hint is not None:
            key = R._workload_key(dev, C, t["means"].shape[0], W, H)
            R._len_hint[key] = hint
            R._seg_sticky.pop(key, None)   # (the stride is sticky per workload: a test that dictates the hint starts afresh)
        before = R.fused_calls[0]
        sp = R.SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], vm, Ks, W, H, want_isect_ids=True,
                                order=order)
        took_fused = R.fused_calls[0] > before
        tl = sp.tl
        n_box, n_isects = tl.n_box, tl.n_isects  # resolves (and rebuilds on overflow)
        return sp, tl, took_fused, n_box, n_isects
    finally:
        R.FUSED_LISTS = old


def _slots(tl, n_box):
    """compact index of every box intersection j <= n_box (the gradient-slot numbering)"""
    ks = tl.keep_scan.cpu().numpy().astype(np.int64).reshape(-1, 2049)
    j = np.arange(n_box + 1)
    return ks[j >> 11, 0] + ks[j >> 11, 1 + (j & 2047)]


def _check_order(tl, nt):
    order = tl.tile_order.cpu().numpy()
    ids = order[order >= 0]
    heavy = ids[(ids & (1 << 30)) != 0] & ~(1 << 30)
    light = ids[(ids & (1 << 30)) == 0]
    assert len(np.unique(heavy)) * 4 == len(heavy)
    assert len(np.unique(light)) == len(light)
    assert sorted(np.unique(heavy).tolist() + light.tolist()) == list(range(nt))


def _compare(a, b, n_box, n_isects, nt):
    assert torch.equal(a.cum_tiles, b.cum_tiles)
    assert torch.equal(a.tile_offsets, b.tile_offsets)
    assert torch.equal(a.flatten_ids[:n_isects], b.flatten_ids[:n_isects])
    assert torch.equal(a.isect_ids[:n_isects], b.isect_ids[:n_isects])
    assert np.array_equal(_slots(a, n_box), _slots(b, n_box))
    _check_order(a, nt)
    _check_order(b, nt)


@pytest.mark.parametrize("culling", [True, False])
@pytest.mark.parametrize("n,W,H,C", [(20_000, 512, 288, 1), (7_000, 333, 201, 3), (300_000, 1352, 1014, 1)])
def test_fused_lists_equal_two_pass_lists(hip_device, culling, n, W, H, C):
    import mobgs_amd.rendering as R
    cam = SynthCamera().scaled(W, H)
    s = splat_inputs(n, cam, 3, 9)
    R.set_tile_culling(culling)
    try:
        _, ref, took, n_box, n_isects = _project(s, hip_device, W, H, fused=False, C=C)
        assert not took
        hint = ref.max_tile_len
        _, got, took, n_box2, n_isects2 = _project(s, hip_device, W, H, fused=True, hint=hint, C=C)
        assert took, "the single-pass path was not taken"
        assert got.rebuilds == 0
        assert (n_box2, n_isects2, got.max_tile_len) == (n_box, n_isects, ref.max_tile_len)
        nt = C * ((W + 15) // 16) * ((H + 15) // 16)
        _compare(got, ref, n_box, n_isects, nt)
    finally:
        R.set_tile_culling(True)


def test_python_host_path_takes_the_fused_entry_point_too(hip_device):
    import mobgs_amd.rendering as R
    from mobgs_amd import _fast
    W, H = 400, 240
    cam = SynthCamera().scaled(W, H)
    s = splat_inputs(15_000, cam, 5, 9)
    _, ref, _, n_box, n_isects = _project(s, hip_device, W, H, fused=False)
    _fast.reset(False)
    try:
        _, got, took, _, _ = _project(s, hip_device, W, H, fused=True, hint=ref.max_tile_len)
    finally:
        _fast.reset(None)
    assert took and got.rebuilds == 0
    _compare(got, ref, n_box, n_isects, ((W + 15) // 16) * ((H + 15) // 16))


def test_segment_overflow_falls_back_to_the_two_pass_rebuild(hip_device):
    """A hint far below the true longest list: the device hands out empty lists, the host sees longest > seg_stride,
    rebuilds with the two-pass entry points and the NEXT frame's segments are large enough."""
    import mobgs_amd.rendering as R
    W, H = 512, 288
    cam = SynthCamera().scaled(W, H)
    s = splat_inputs(40_000, cam, 7, 9)
    _, ref, _, n_box, n_isects = _project(s, hip_device, W, H, fused=False)
    assert ref.max_tile_len > 100
    over = R.seg_overflows[0]
    _, got, took, _, _ = _project(s, hip_device, W, H, fused=True, hint=20)  # segments of 64 keys
    assert took and got.rebuilds == 1 and R.seg_overflows[0] == over + 1
    nt = ((W + 15) // 16) * ((H + 15) // 16)
    _compare(got, ref, n_box, n_isects, nt)
    _, again, took, _, _ = _project(s, hip_device, W, H, fused=True)  # the hint the rebuild left behind
    assert took and again.rebuilds == 0
    _compare(again, ref, n_box, n_isects, nt)


def test_render_is_bit_identical_with_fused_lists(hip_device):
    """End to end: the lean render() forward + backward through single-pass lists equals the two-pass run bit for bit
    (same lists, same slots -> same kernels on the same data)."""
    import mobgs_amd.rendering as R
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.gaussian_renderer import render
    from mobgs_amd.helper_model import Sandwich
    from mobgs_amd.synth import dynamic_extras, gaussian_cloud
    dev = hip_device
    W, H = 640, 360
    scam = SynthCamera().scaled(W, H)
    stat_p, dyn_p = gaussian_cloud(30_000, scam, 0), gaussian_cloud(15_000, scam, 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], 0)
    torch.manual_seed(0)
    dec = Sandwich(9, 3).to(dev)
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    # ONE scene object for the three renders: list-length hints are kept per scene (rendering.hint_scope, round 6), and the
    # single-pass path needs the hint its own previous frame left
    stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
    dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
    for fused in (False, True, True):
        R.FUSED_LISTS = fused
        try:
            for p_ in (stat._xyz, dyn.control_xyz, stat._opacity):
                p_.grad = None
            cam = PinholeCamera(W, H, scam.K, torch.eye(4), scam.time, scam.max_time, device=dev)
            before = R.fused_calls[0]
            out = render(cam, stat, dyn, None, torch.zeros(9, device=dev))
            ((out["render"] * v).sum() + out["depth"].sum()).backward()
            res[(fused, R.fused_calls[0] > before)] = (out["render"].detach().clone(), out["depth"].detach().clone(),
                                                      stat._xyz.grad.clone(), dyn.control_xyz.grad.clone(),
                                                      stat._opacity.grad.clone())
        finally:
            R.FUSED_LISTS = True
    assert (True, True) in res, "the single-pass path never ran"
    for a, b in zip(res[(False, False)], res[(True, True)]):
        assert torch.equal(a, b)


@pytest.mark.parametrize("n,W,H,C", [(20_000, 512, 288, 1), (7_000, 333, 201, 3), (300_000, 1352, 1014, 1)])
def test_enumeration_order_changes_nothing_but_the_slot_numbering(hip_device, n, W, H, C):
    """SharedProjection(order=): the bounding-box intersections enumerated along a Morton curve of the positions (and, as
    a stress case, in a random order).  The lists -- contents, order, offsets -- must equal the unordered build's; the
    gradient-slot numbering changes, but it must stay a bijection onto 0 .. I-1 in which every splat's slots are
    consecutive and in its own tile order (cum_tiles[g] .. cum_tiles[g] + tiles_per_gauss[g])."""
    import mobgs_amd.rendering as R
    cam = SynthCamera().scaled(W, H)
    s = splat_inputs(n, cam, 3, 9)
    _, ref, _, n_box, n_isects = _project(s, hip_device, W, H, fused=False, C=C)
    morton = R.spatial_order(s["means"].to(hip_device), C)
    rnd = torch.randperm(C * n, generator=torch.Generator().manual_seed(5)).to(torch.int32).to(hip_device)
    assert sorted(morton.cpu().tolist()) == list(range(C * n))
    for name, order in (("morton", morton), ("random", rnd)):
        _, got, took, n_box2, n_isects2 = _project(s, hip_device, W, H, fused=True, hint=ref.max_tile_len, C=C, order=order)
        assert took and got.rebuilds == 0 and (n_box2, n_isects2) == (n_box, n_isects), name
        assert torch.equal(got.tile_offsets, ref.tile_offsets), name
        assert torch.equal(got.flatten_ids[:n_isects], ref.flatten_ids[:n_isects]), name
        assert torch.equal(got.isect_ids[:n_isects], ref.isect_ids[:n_isects]), name
        tpg = got.tiles_per_gauss.reshape(-1).cpu().numpy().astype(np.int64)
        cum = got.cum_tiles.cpu().numpy().astype(np.int64)
        assert cum[C * n] == n_box
        # the splats' intervals [cum[g], cum[g] + tpg[g]) tile 0 .. n_box exactly, in the order given
        o = order.cpu().numpy().astype(np.int64)
        assert np.array_equal(cum[o], np.concatenate([[0], np.cumsum(tpg[o])[:-1]])), name
        slots = _slots(got, n_box)
        ref_slots, ref_cum = _slots(ref, n_box), ref.cum_tiles.cpu().numpy().astype(np.int64)
        # per splat: the same number of kept intersections, at the same positions inside its box, as the unordered build
        vis = np.nonzero(tpg)[0]
        pick = vis[:: max(1, len(vis) // 4000)]
        for g in pick:
            a = slots[cum[g]: cum[g] + tpg[g] + 1] - slots[cum[g]]
            b = ref_slots[ref_cum[g]: ref_cum[g] + tpg[g] + 1] - ref_slots[ref_cum[g]]
            assert np.array_equal(a, b), (name, int(g))
        assert slots[n_box] == n_isects, name


def test_render_is_bit_identical_with_an_enumeration_order(hip_device):
    """render() forward + backward with the cached Morton order (the default) against MOBGS_ENUM_ORDER off: every output
    and gradient bit for bit -- the per-splat slot order, hence the summation order of the slot reduction, is unchanged."""
    import mobgs_amd.gaussian_renderer as GR
    import mobgs_amd.rendering as R
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.helper_model import Sandwich
    from mobgs_amd.synth import dynamic_extras, gaussian_cloud
    dev = hip_device
    W, H = 640, 360
    scam = SynthCamera().scaled(W, H)
    stat_p, dyn_p = gaussian_cloud(30_000, scam, 0), gaussian_cloud(15_000, scam, 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], 0)
    torch.manual_seed(0)
    dec = Sandwich(9, 3).to(dev)
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res = {}
    for use_order in (False, True):
        GR.ENUM_ORDER = use_order
        try:
            stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
            dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
            cam = PinholeCamera(W, H, scam.K, torch.eye(4), scam.time, scam.max_time, device=dev)
            for rep in range(2):   # (the first call of a workload has no list-length hint yet: two-pass path, no order)
                for p in (stat._xyz, dyn.control_xyz, stat._opacity, dyn._features_dc):
                    p.grad = None
                before = R.fused_calls[0]
                out = GR.render(cam, stat, dyn, None, torch.zeros(9, device=dev), get_static=True, get_dynamic=True)
                ((out["render"] * v).sum() + out["depth"].sum() + out["d_alpha"].sum() + out["s_render"].sum()).backward()
            assert R.fused_calls[0] > before
            res[use_order] = (out["render"].detach().clone(), out["depth"].detach().clone(), out["d_alpha"].detach().clone(),
                              stat._xyz.grad.clone(), dyn.control_xyz.grad.clone(), stat._opacity.grad.clone(),
                              dyn._features_dc.grad.clone(), out["viewspace_points"].grad.clone())
        finally:
            GR.ENUM_ORDER = True
    for a, b in zip(res[False], res[True]):
        assert torch.equal(a, b)


def test_rows_stored_in_morton_order_render_the_same_scene(hip_device):
    """GaussianParams.spatial_sort_(): the rows of both sets permuted along a Morton curve, the renderer told so
    (rendering.COHERENT: LDS-ranked binning without an order's indirection).  The image is the same scene's -- the
    per-pixel blend order is the depth order either way (equal depths aside), so it agrees to the last bits -- and row r
    of every gradient is row order[r] of the unsorted model's, to summation order."""
    import mobgs_amd.gaussian_renderer as GR
    import mobgs_amd.rendering as R
    from helpers import close
    from mobgs_amd.camera import PinholeCamera
    from mobgs_amd.gaussian_model import GaussianParams
    from mobgs_amd.helper_model import Sandwich
    from mobgs_amd.synth import dynamic_extras, gaussian_cloud
    dev = hip_device
    W, H = 640, 360
    scam = SynthCamera().scaled(W, H)
    stat_p, dyn_p = gaussian_cloud(30_000, scam, 0), gaussian_cloud(15_000, scam, 1)
    dyn_x = dynamic_extras(dyn_p["xyz"], 0)
    torch.manual_seed(0)
    dec = Sandwich(9, 3).to(dev)
    v = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    res, orders = {}, None
    seen = []
    real = GR._enum_order

    def spy(*a, **k):
        seen.append(real(*a, **k))
        return seen[-1]
    GR._enum_order = spy
    try:
        for sort in (False, True):
            stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
            dyn = GaussianParams(dyn_p, dyn_x, dec, dev, requires_grad=True)
            if sort:
                orders = (stat.spatial_sort_(), dyn.spatial_sort_())
                assert stat.rows_coherent == 30_000 and dyn.rows_coherent == 15_000
            cam = PinholeCamera(W, H, scam.K, torch.eye(4), scam.time, scam.max_time, device=dev)
            for rep in range(2):
                for p in (stat._xyz, dyn.control_xyz, stat._opacity, dyn._features_dc):
                    p.grad = None
                out = GR.render(cam, stat, dyn, None, torch.zeros(9, device=dev))
                ((out["render"] * v).sum() + out["depth"].sum()).backward()
            res[sort] = (out["render"].detach().clone(), out["depth"].detach().clone(), stat._xyz.grad.clone(),
                         stat._opacity.grad.clone(), dyn.control_xyz.grad.clone(), dyn._features_dc.grad.clone(),
                         out["radii"].clone())
            assert (seen[-1] == R.COHERENT) if sort else torch.is_tensor(seen[-1])
    finally:
        GR._enum_order = real
    so, do = orders
    a, b = res[False], res[True]
    close(b[0], a[0], 0, 2e-6, "image, sorted rows")
    close(b[1], a[1], 0, 2e-5 * float(a[1].abs().max()), "depth, sorted rows")
    n_s = so.numel()
    assert torch.equal(b[6][:n_s], a[6][so]) and torch.equal(b[6][n_s:], a[6][n_s:][do])   # radii: exact, row for row
    for i, o in ((2, so), (3, so), (4, do), (5, do)):
        close(b[i], a[i][o], 2e-5, 2e-5 * float(a[i].abs().max()), f"gradient {i}, sorted rows")
    # the sort is idempotent up to ties and a second render of the sorted model takes the same path
    stat = GaussianParams(stat_p, None, dec, dev, requires_grad=True)
    o1 = stat.spatial_sort_()
    o2 = stat.spatial_sort_()
    pos = stat._xyz.detach()
    assert torch.equal(pos[o2], pos) or float((pos[o2] - pos).abs().max()) < 1e-2   # (equal codes may swap)
    assert o1.numel() == 30_000


def test_trainable_table_sorts_parameters_moments_and_statistics_together(hip_device):
    """densify.TrainableGaussians.spatial_sort_(): one gather moves parameters, Adam moments and densification
    statistics; keep_sorted re-sorts after a densification; pruning keeps the order."""
    from types import SimpleNamespace
    from mobgs_amd.densify import TrainableGaussians
    from mobgs_amd.synth import dynamic_extras, gaussian_cloud
    dev = hip_device
    scam = SynthCamera().scaled(320, 200)
    p = gaussian_cloud(5_000, scam, 3)
    g = TrainableGaussians(p, dynamic_extras(p["xyz"], 3), device=dev)
    opt = SimpleNamespace(position_lr_init=1e-4, feature_lr=1e-3, featuret_lr=1e-3, opacity_lr=1e-2, scaling_lr=1e-3,
                          rotation_lr=1e-3, omega_lr=1e-3, zeta_lr=1e-3, trbfc_lr=1e-3, trbfs_lr=1e-3, movelr=1.0,
                          rgb_lr=1e-3, percent_dense=0.01)
    try:
        g.training_setup(opt)
    except Exception as e:  # (an options object with other field names: the table alone is what this test is about)
        pytest.skip(f"training_setup needs other option fields: {e}")
    for grp in g.optimizer.param_groups:
        for q in grp["params"]:
            if q.requires_grad:
                q.grad = torch.randn_like(q)
    g.optimizer.step()
    g.max_radii2D.copy_(torch.arange(g._n, device=dev, dtype=torch.float32))
    xyz, m1 = g._xyz.detach().clone(), None
    for grp in g.optimizer.param_groups:
        if grp["name"] == "xyz":
            m1 = g.optimizer.state[grp["params"][0]]["exp_avg"].clone()
    order = g.spatial_sort_()
    assert g.rows_coherent == g._n
    assert torch.equal(g._xyz.detach(), xyz[order])
    assert torch.equal(g.max_radii2D, order.to(torch.float32))
    for grp in g.optimizer.param_groups:
        if grp["name"] == "xyz":
            assert torch.equal(g.optimizer.state[grp["params"][0]]["exp_avg"], m1[order])
    mask = torch.zeros(g._n, dtype=torch.bool, device=dev)
    mask[::7] = True
    g.prune_points(mask)
    assert g.rows_coherent == g._n
    g.keep_sorted = True
    g.densify_and_clone(torch.rand(g._n, 1, device=dev), 0.5, 10.0)
    assert g.rows_coherent == g._n
    g.keep_sorted = False
    g.densify_and_clone(torch.rand(g._n, 1, device=dev), 0.5, 10.0)
    assert g.rows_coherent == -1
