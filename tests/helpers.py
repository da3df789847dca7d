"""Shared test helpers: rebuild scenes from golden fixtures, tolerant comparisons."""
import math
import os

import numpy as np
import torch

from mobgs_amd.camera import PinholeCamera
from mobgs_amd.gaussian_model import GaussianParams
from mobgs_amd.helper_model import Sandwich

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STAT_KEYS = ["xyz", "scaling", "rotation", "opacity", "features_dc", "features_t"]
DYN_KEYS = ["omega", "trbf_center", "control_xyz", "current_control_num"]


def load(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


def scene_from_fixture(fx, device="cpu", requires_grad=True):
    """-> cam, stat_pc, dyn_pc, bg, w2c (all on `device`)."""
    T = torch.from_numpy
    dec = Sandwich(9, 3)
    with torch.no_grad():
        dec.mlp1.weight.copy_(T(fx["in_w1"]))
        dec.mlp2.weight.copy_(T(fx["in_w2"]))
    dec = dec.to(device)
    for p in dec.parameters():
        p.requires_grad_(requires_grad)
    stat = GaussianParams({k: T(fx["in_s_" + k]) for k in STAT_KEYS}, None, dec, device, requires_grad)
    dyn = GaussianParams({k: T(fx["in_d_" + k]) for k in STAT_KEYS}, {k: T(fx["in_d_" + k]) for k in DYN_KEYS}, dec,
                         device, requires_grad)
    W, H, time, max_time = fx["in_cam"]
    w2c = T(fx["in_w2c"]).to(device)
    cam = PinholeCamera(int(W), int(H), T(fx["in_K"]), w2c, time=float(time), max_time=int(max_time), device=device)
    bg = T(fx["in_bg"]).to(device)
    return cam, stat, dyn, bg, w2c


def leaf_map(stat, dyn):
    d = {"s_" + k: getattr(stat, k) for k in ("_xyz", "_scaling", "_rotation", "_opacity", "_features_dc",
                                               "_features_t")}
    d.update({"d_" + k: getattr(dyn, k) for k in ("_scaling", "_rotation", "_opacity", "_features_dc",
                                                  "_features_t", "_omega", "control_xyz")})
    d["w1"] = dyn.rgbdecoder.mlp1.weight
    d["w2"] = dyn.rgbdecoder.mlp2.weight
    return d


def render_loss(out, fx, device="cpu"):
    """The scalar the fixture generator back-propagated (tests/golden/make_golden.py gen_render)."""
    v_render = torch.from_numpy(fx["cot_v_render"]).to(device)
    v_depth = torch.from_numpy(fx["cot_v_depth"]).to(device)
    get_static, get_dynamic = bool(fx["opt"][0]), bool(fx["opt"][1])
    loss = (out["render"] * v_render).sum() + (out["depth"] * v_depth).sum()
    if get_static:
        loss = loss + (out["s_render"] * v_render).sum() * 0.5 + (out["s_alpha"] * v_depth).sum() * 0.25
    if get_dynamic:
        loss = loss + (out["d_render"] * v_render).sum() * 0.5 + (out["d_alpha"] * v_depth).sum() * 0.25 \
            + (out["d_depth"] * v_depth).sum() * 0.125
    return loss


def observe(what, nbad, numel, max_err, flip_frac, flip_atol, ref_max):
    """MOBGS_TEST_REPORT=1 (with pytest -s): one line per comparison with what was OBSERVED next to what is ALLOWED --
    the tolerances in the tests are set from these (at most ~3x the observed figures)."""
    if os.environ.get("MOBGS_TEST_REPORT") == "1":
        import inspect
        site = "?"
        for fr in inspect.stack()[1:]:
            base = os.path.basename(fr.filename)
            if base.startswith("test_") and fr.function not in ("close", "_close", "observe"):
                site = f"{base}:{fr.lineno}"
                break
        print(f"[obs] <{site}> {what}: beyond-tight {nbad}/{numel} = {nbad / max(1, numel):.2e} (allowed {flip_frac:.1e}); "
              f"max err {max_err:.3e} = {max_err / max(ref_max, 1e-30):.2e} of ref max (flip allowance {flip_atol:.3e})")


MIN_ROUND_UP = 256  # tensors with at least this many elements get their flip allowance rounded up (see close)


def close(a, b, rtol, atol, what, flip_frac=0.0, flip_atol=0.0):
    """|a-b| <= atol + rtol*|b|, except that a fraction `flip_frac` of the elements may be off by up to
    `flip_atol` (alpha-threshold / transmittance-stop decisions that flip with the last bit of exp()).  For per-splat
    and per-pixel tensors (>= MIN_ROUND_UP elements) the count is rounded UP: `flip_frac` is a rate per compared element,
    and an element of a per-splat gradient aggregates the hundreds of (pixel, splat) pairs of that splat -- a tensor with
    fewer than 1 / flip_frac elements can still hold one that a flipped pair has touched (e.g. the 450 opacity gradients
    of tests/golden/get_flow_grad.npz).  SMALL tensors (the 16 entries of a camera-matrix gradient, a handful of weights)
    are sums over the whole image: one flipped pair moves them by nothing visible, so they get no free element (the
    count is rounded DOWN)."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    nbad = int(bad.sum())
    msg = f"{what}: {nbad}/{bad.numel()} off, max err {float(err.max()) if err.numel() else 0:.3e} " \
          f"(ref max {float(b.abs().max()) if b.numel() else 0:.3e})"
    observe(what, nbad, bad.numel(), float(err.max()) if err.numel() else 0.0, flip_frac, flip_atol,
            float(b.abs().max()) if b.numel() else 0.0)
    allowed = math.ceil(flip_frac * bad.numel()) if bad.numel() >= MIN_ROUND_UP else math.floor(flip_frac * bad.numel())
    assert nbad <= allowed, msg
    if nbad:
        assert float(err.max()) <= flip_atol, msg


def psnr(img, target):
    mse = ((img.double() - target.double()) ** 2).mean()
    return float(20 * torch.log10(1.0 / torch.sqrt(mse)))


def mid_planes(shapes, seed):
    """Plane values of tests/golden/deform_mid.npz (same draws as make_golden.mid_planes: the fixture does not store
    the 2.1 M plane values, only the seed)."""
    g = torch.Generator().manual_seed(int(seed))
    return [0.5 + 0.5 * torch.rand(tuple(s), generator=g) for s in shapes]


def check_deform_fixture_grads(fx, get_plane_grad, what=""):
    """Plane gradients of the mid-size fixture: time planes in full, spatial planes by {sum, sum|.|, 2048 samples}.
    One point whose ReLU decision flips (see test_gpu_config3) moves the 4 x 32 entries it touches in every plane
    by a few per cent of its own contribution: up to 2 % of a small plane's entries may be off by 2 % of the max."""
    for li in range(3):
        for pi in range(6):
            g = get_plane_grad(li, pi).detach().cpu()
            if pi in (2, 4, 5):
                ref = fx[f"gplane_{li}_{pi}"]
                sc = float(np.abs(ref).max())
                close(g, ref, 1e-3, 1e-4 * sc + 1e-6, f"{what}grad plane {li}.{pi}", flip_frac=0.02, flip_atol=0.02 * sc)
            else:
                flat = g.reshape(-1)
                ref = fx[f"gplane_val_{li}_{pi}"]
                s_ref, a_ref = fx[f"gplane_sum_{li}_{pi}"]
                sc = float(np.abs(ref).max())
                close(flat[torch.from_numpy(fx[f"gplane_idx_{li}_{pi}"])], ref, 1e-3, 1e-4 * sc + 1e-6,
                      f"{what}grad plane {li}.{pi} samples", flip_frac=0.02, flip_atol=0.02 * sc)
                assert abs(float(flat.double().abs().sum()) - a_ref) <= 1e-4 * a_ref, (li, pi)
                assert abs(float(flat.double().sum()) - s_ref) <= 1e-4 * a_ref, (li, pi)


def close_image_with_blend_flips(img, ref, alphas_ref, colors_absmax, depth_spread, tight_atol, what, flip_frac=2e-3,
                                 n_colour_channels=None, alphas_img=None):
    """Image comparison whose discrete-decision allowance is DERIVED, not guessed.  Two fp32 implementations of
    alpha = o exp(-sigma) disagree in the last bit, so a splat sitting on the 1/255 skip threshold is blended by one
    and skipped by the other.  The pixel then moves by at most w (|c| + |pixel|) with w = alpha T <= 1/255 in a colour
    channel, and an expected-depth channel (accumulated depth / alpha) by at most w spread / alpha_pixel.  `flip_frac`
    of the elements may use that bound (x 2: the flip can also cascade into the next splat's weight), everything else
    must meet `tight_atol`.
    The OTHER discrete decision is the transmittance stop: T (1 - alpha) <= 1e-4 ends a pixel's walk WITHOUT blending
    the splat, so its two outcomes are "T stays" and "T drops to ~1e-4" -- in front of an opaque splat (alpha up to the
    0.999 clamp) a step of alpha T with T up to 0.1, not 1e-4 (found by scripts/soak_parity.py, seed 4203 case 48).
    With `alphas_img` (the coverage of `img`) such pixels are recognised -- one of the two results sits at the stop
    threshold -- and bounded by the other one's transmittance: max(T_ref, T_img) (|c| + |pixel|).
    -> (number of flipped elements, their largest error): logged by the callers."""
    img = torch.as_tensor(img).detach().cpu().double()
    ref = torch.as_tensor(ref).detach().cpu().double()
    assert img.shape == ref.shape, f"{what}: shape {tuple(img.shape)} vs {tuple(ref.shape)}"
    a = torch.as_tensor(alphas_ref).detach().cpu().double().reshape(ref.shape[:-1] + (1,))
    C = ref.shape[-1]
    nc = C if n_colour_channels is None else n_colour_channels
    w = 1.0 / 255.0 * 1.001
    bound = torch.empty_like(ref)
    bound[..., :nc] = 2.0 * w * (colors_absmax + ref[..., :nc].abs())
    if nc < C:
        bound[..., nc:] = 2.0 * w * depth_spread / a.clamp_min(1.0 / 255.0)
    if alphas_img is not None:
        ai = torch.as_tensor(alphas_img).detach().cpu().double().reshape(ref.shape[:-1] + (1,))
        t_ref, t_img = 1.0 - a, 1.0 - ai
        at_stop = torch.minimum(t_ref, t_img) <= 1.2e-4
        step = 1.001 * torch.maximum(t_ref, t_img)
        stop_bound = torch.empty_like(ref)
        stop_bound[..., :nc] = step * (colors_absmax + ref[..., :nc].abs())
        if nc < C:
            stop_bound[..., nc:] = step * depth_spread / torch.minimum(a, ai).clamp_min(1.0 / 255.0)
        bound = torch.where(at_stop.expand_as(ref), torch.maximum(bound, stop_bound), bound)
    err = (img - ref).abs()
    bad = err > tight_atol
    nbad = int(bad.sum())
    msg = f"{what}: {nbad}/{bad.numel()} beyond {tight_atol:.1e}, max err {float(err.max()):.3e}"
    observe(what + " (blend-flip bound)", nbad, bad.numel(), float(err.max()), flip_frac, float(bound.max()),
            float(ref.abs().max()))
    assert nbad <= flip_frac * bad.numel(), msg
    over = bad & (err > bound)
    assert not bool(over.any()), msg + f"; {int(over.sum())} exceed the one-blend-step bound, worst " \
        f"{float((err / bound)[over].max()):.2f}x"
    return nbad, (float(err[bad].max()) if nbad else 0.0)


def decoded_flip_bound(decoder, colors_absmax):
    """What ONE flipped blend decision can do to a DECODED colour (render / s_render / d_render and their K-sub-frame
    mean): the flipped splat has weight w = alpha T <= 1/255, a composited feature moves by at most 2 w (|c| + |pixel|)
    <= 2 w 2 cmax (x 2: the flip also changes the next weights), and rgb = sigmoid(albedo + W2 relu(W1 .)) passes that on
    with the decoder's Lipschitz factor 1/4 (1 + |W2|_inf |W1|_inf).  Derived from the decoder's own weights and the
    splat colours' range -- not a flat fraction of the image range."""
    w = 1.001 / 255.0
    w1 = decoder.mlp1.weight.detach().reshape(6, 12).cpu()
    w2 = decoder.mlp2.weight.detach().reshape(3, 6).cpu()
    lip = 0.25 * (1.0 + float(w2.abs().sum(1).max()) * float(w1.abs().sum(1).max()))
    return lip * 2.0 * w * 2.0 * float(colors_absmax)


def flow_flip_bound(ref_map):
    """One-blend-step bound for a splatted 2-D flow map given as absolute coordinates `pixel + flow` [..,H,W,2]
    (get_flow()'s exp2mid / mid2exp, /root/reference/gaussian_renderer/__init__.py:436-476): a splat that flips at the
    1/255 threshold has weight w = alpha T <= 1/255 and moves the splatted flow by at most w (|flow_splat| + |flow_pixel|)
    -- x 2 for the cascade into the next splat's weight.  |flow| is bounded here by the largest flow the map itself
    shows (+ 1 px of slack for the splats' own spread around it)."""
    r = torch.as_tensor(ref_map).detach().cpu().double()
    H, W = r.shape[-3], r.shape[-2]
    gx = torch.arange(W, dtype=torch.float64)[None, :].expand(H, W)
    gy = torch.arange(H, dtype=torch.float64)[:, None].expand(H, W)
    flow = r.reshape(-1, H, W, 2) - torch.stack([gx, gy], -1)
    fmax = float(flow.abs().max()) + 1.0
    return 2.0 * (1.001 / 255.0) * 2.0 * fmax


def close_point_rows(a, b, rtol, atol, what, flip_rows=0.0, share=1.0):
    """Per-point tensors [N, k] (gradients w.r.t. positions / scales / rotations of the deformation network's inputs).
    A hidden unit whose pre-activation is zero to rounding takes the other ReLU branch, and a coordinate within rounding
    of a grid line takes the neighbouring bilinear cell's slope: such a decision changes THAT point's row, by at most a
    fraction `share` of the row's own magnitude, and no other row.  So: rows with an element beyond the tight tolerance
    are counted (<= flip_rows x N) and each of them is bounded by share x (its own largest |ref| entry) -- instead of
    a flat fraction of the tensor's maximum."""
    a = torch.as_tensor(a).detach().cpu().double()
    b = torch.as_tensor(b).detach().cpu().double()
    assert a.shape == b.shape and a.dim() == 2, f"{what}: shapes {tuple(a.shape)} vs {tuple(b.shape)}"
    err = (a - b).abs()
    bad = err > atol + rtol * b.abs()
    rows = bad.any(1)
    nrows = int(rows.sum())
    own = b.abs().max(1).values
    worst = float((err.max(1).values[rows] / (own[rows] + atol)).max()) if nrows else 0.0
    observe(what + " (rows)", nrows, a.shape[0], float(err.max()), flip_rows, share, float(b.abs().max()))
    msg = f"{what}: {nrows}/{a.shape[0]} rows off (allowed {flip_rows:.1e}), worst {worst:.2f} x the row's own magnitude"
    assert nrows <= flip_rows * a.shape[0], msg
    if nrows:
        assert worst <= share, msg
