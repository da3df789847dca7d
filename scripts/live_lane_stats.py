"""Live-lane statistics of the compositors on the benchmark lists (measure before rebuilding).

For every (tile, splat) entry of the headline workload's lists (BASELINE config #2: 200 k + 100 k splats, 1352x1014) and
every pixel of its 16x16 tile, decide whether the pair is LIVE:
    forward  : alpha >= 1/255 (and sigma >= 0) and the pixel had not stopped before this entry (T_before > 1e-4 rule)
    backward : the pair was blended (forward-live and not the stopping entry itself)
and count, for several block shapes a SIMD program could branch on (a block is evaluated as soon as ONE of its pixels
is live), how many lane-evaluations a compositor of that granularity has to issue:
    16x16 (whole tile, no sub-tile culling) | 8x8 quadrant (the round-2 kernels) | 8x4 / 4x8 (half wave) |
    4x4 / 8x2 / 16x1 (one 16-lane DPP row) | 2x2 (quad) | 1x1 (= the live pairs themselves)
plus the histogram of live lanes per evaluated (entry, 8x8 quadrant) pair and the per-tile list statistics a
row-granular walk would see (sum over the 16 4x4 blocks of a tile of their list lengths, their maximum, ...).

    python scripts/live_lane_stats.py [--out profiles/r03/live_lane_stats.json]
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import build_scene  # noqa: E402
from mobgs_amd.rendering import rasterization  # noqa: E402
from oracle import render_torch as R  # noqa: E402  (activation glue only; statistics script, not the product path)

ap = argparse.ArgumentParser()
ap.add_argument("--out", default=None)
ap.add_argument("--ns", type=int, default=200_000)
ap.add_argument("--nd", type=int, default=100_000)
ap.add_argument("--width", type=int, default=1352)
ap.add_argument("--height", type=int, default=1014)
args = ap.parse_args()

dev = torch.device("cuda")
scam, cam, stat, dyn, (stat_p, dyn_p, dyn_x) = build_scene(dev, args.ns, args.nd, args.width, args.height)
W, H = args.width, args.height
# the activated splats of the benchmark scene, as bench.cpu_baseline() builds them
ctrl = R.hermite(dyn_x["control_xyz"], torch.tensor(scam.time), dyn_x["current_control_num"]) * 1e-2
tfp = scam.time - dyn_x["trbf_center"]
means = torch.cat([stat_p["xyz"], ctrl]).to(dev)
quats = torch.cat([stat_p["rotation"], dyn_p["rotation"] + tfp * dyn_x["omega"]]).to(dev)
scales = torch.exp(torch.cat([stat_p["scaling"], dyn_p["scaling"]])).to(dev)
opac = torch.sigmoid(torch.cat([stat_p["opacity"], dyn_p["opacity"]])).squeeze(-1).to(dev)
cols = torch.cat([torch.cat([stat_p["features_dc"], 0 * stat_p["features_t"]], 1),
                  torch.cat([dyn_p["features_dc"], tfp * dyn_p["features_t"]], 1)]).to(dev)
viewmats = torch.eye(4, device=dev)[None]
Ks = scam.K.to(dev)[None]
with torch.no_grad():
    img, alpha_img, meta = rasterization(means, quats, scales, opac, cols, viewmats, Ks, W, H, packed=False,
                                         backgrounds=torch.zeros(1, 9, device=dev), render_mode="RGB+ED")
ids = meta["flatten_ids"].long()
offs = meta["isect_offsets"].reshape(-1).long()
tw = int(meta["tile_width"])
nt = offs.numel()
I = ids.numel()
ends = torch.cat([offs[1:], torch.tensor([I], device=dev)])
lens = ends - offs
tile_of = torch.repeat_interleave(torch.arange(nt, device=dev), lens)
m2 = meta["means2d"].reshape(-1, 2)[ids]
con = meta["conics"].reshape(-1, 3)[ids]
op = opac[ids]
ty, tx = tile_of // tw, tile_of % tw
lx = torch.arange(16, device=dev) + 0.5

# pass 1: alpha per (entry, pixel), segmented transmittance per pixel along each tile's list
# (float64 log-domain cumulative sums; statistics only -- a pixel whose T sits on the 1e-4 edge may be
# classified differently from the kernel, a handful in 10^8)
logq = torch.empty(I, 256, dtype=torch.float32, device=dev)   # log(1 - alpha) where the alpha test passes, else 0
passed = torch.empty(I, 256, dtype=torch.bool, device=dev)
CH = 200_000
for c0 in range(0, I, CH):
    sl = slice(c0, min(I, c0 + CH))
    px = (tx[sl] * 16)[:, None, None] + lx[None, None, :]
    py = (ty[sl] * 16)[:, None, None] + lx[None, :, None]
    dx = m2[sl, 0][:, None, None] - px
    dy = m2[sl, 1][:, None, None] - py
    sig = 0.5 * (con[sl, 0][:, None, None] * dx * dx + con[sl, 2][:, None, None] * dy * dy) \
        + con[sl, 1][:, None, None] * dx * dy
    al = torch.clamp(op[sl][:, None, None] * torch.exp(-sig), max=0.999)
    ok = (sig >= 0) & (al >= 1 / 255) & (px < W) & (py < H)
    passed[sl] = ok.reshape(-1, 256)
    logq[sl] = torch.where(ok, torch.log1p(-al), torch.zeros_like(al)).reshape(-1, 256)
cum = torch.cumsum(logq.double(), 0)                                  # inclusive, over the concatenated lists
start_cum = torch.zeros(nt, 256, dtype=torch.float64, device=dev)
nz = offs > 0
start_cum[nz] = cum[offs[nz] - 1]                                      # value just before each tile's first entry
LOG_STOP = torch.log(torch.tensor(1e-4, dtype=torch.float64, device=dev))
live_fwd = torch.empty_like(passed)
live_bwd = torch.empty_like(passed)
for c0 in range(0, I, CH):
    sl = slice(c0, min(I, c0 + CH))
    t_after = cum[sl] - start_cum[tile_of[sl]]
    t_before = t_after - logq[sl].double()
    running = t_before > LOG_STOP                                       # the pixel had not stopped yet
    live_fwd[sl] = passed[sl] & running
    live_bwd[sl] = passed[sl] & (t_after > LOG_STOP)
del cum, logq


def block_any(live, by, bx):
    """[I, 16/by, 16/bx] bool: the (by x bx)-pixel block holds a live pixel."""
    v = live.reshape(-1, 16 // by, by, 16 // bx, bx)
    return v.any(dim=4).any(dim=2)


def workers(live, b4):
    """Steps of a wave whose W workers (16 column workers on 4x4 blocks / 64 lanes on 2x2 blocks) walk their own
    compacted lists: per synchronisation unit (batch of 64 / 128 entries, or the whole tile) the slowest worker."""
    res = {}
    first = offs[tile_of]
    pos = torch.arange(I, device=dev) - first
    nb64 = (lens + 63) // 64
    for name_w, blk in (("16x(4x4)", b4), ("64x(2x2)", block_any(live, 2, 2).reshape(-1, 64))):
        Wn = blk.shape[1]
        r = {"pairs": int(blk.sum().item())}
        for B in (64, 128, 256):
            nb = (lens + B - 1) // B
            base = torch.cumsum(nb, 0) - nb
            bid = base[tile_of] + pos // B
            cnt = torch.zeros(int(nb.sum().item()), Wn, dtype=torch.long, device=dev)
            cnt.index_add_(0, bid, blk.long())
            r[f"steps_sync{B}"] = int(cnt.max(1).values.sum().item())
            if B == 64:
                # run-ahead: with NB batch buffers a worker that finished batch n goes on into batch n + 1 as soon as it
                # is staged, and batch n + 1 is staged once batch n + 1 - NB is complete (all workers past it)
                nbmax = int(nb.max().item())
                dense = torch.zeros(nt, nbmax, Wn, dtype=torch.long, device=dev)
                tix = tile_of.new_tensor(torch.arange(nt, device=dev)).repeat_interleave(nb)
                bix = torch.arange(int(nb.sum().item()), device=dev) - base.repeat_interleave(nb)
                dense[tix, bix] = cnt
                for NB in (2, 3, 4):
                    fin = torch.zeros(nt, Wn, dtype=torch.long, device=dev)      # f_w[n - 1]
                    done_at = []                                                   # F[n]
                    for n_ in range(nbmax):
                        staged = done_at[n_ - NB] if n_ - NB >= 0 else torch.zeros(nt, dtype=torch.long, device=dev)
                        start = torch.maximum(fin, staged[:, None])
                        fin = start + dense[:, n_]
                        done_at.append(fin.max(1).values)
                    r[f"steps_runahead_{NB}buffers"] = int(done_at[-1].sum().item()) if done_at else 0
        pt = torch.zeros(nt, Wn, dtype=torch.long, device=dev)
        pt.index_add_(0, tile_of, blk.long())
        r["steps_tile"] = int(pt.max(1).values.sum().item())
        r["steps_ideal"] = int(((pt.sum(1) + Wn - 1) // Wn).sum().item())
        res[name_w] = r
    return res


out = {"workload": f"{args.ns}+{args.nd} splats, {W}x{H}", "entries": I, "tiles": nt,
       "pixel_pairs_total": I * 256}
shapes = [(16, 16), (8, 8), (4, 8), (8, 4), (4, 4), (2, 8), (1, 16), (2, 2), (1, 1)]
for name, live in (("alpha_only", passed), ("forward", live_fwd), ("backward", live_bwd)):
    L = int(live.sum().item())
    rows = {}
    for by, bx in shapes:
        nb = int(block_any(live, by, bx).sum().item())
        rows[f"{by}x{bx}"] = {"blocks": nb, "lane_evals": nb * by * bx, "live_fraction": L / max(1, nb * by * bx)}
    q = block_any(live, 8, 8).reshape(-1, 4)
    per_q = live.reshape(-1, 2, 8, 2, 8).permute(0, 1, 3, 2, 4).reshape(-1, 4, 64).sum(-1)
    hist = torch.bincount(per_q[q].clamp(max=64), minlength=65)
    edges = [1, 9, 17, 25, 33, 41, 49, 57, 65]
    hist8 = [int(hist[edges[i]:edges[i + 1]].sum().item()) for i in range(8)]
    nq = q.sum(1)
    # per-tile list lengths a 4x4-block-granular walk would see
    b4 = block_any(live, 4, 4).reshape(-1, 16)                          # [I, 16]
    per_tile = torch.zeros(nt, 16, dtype=torch.long, device=dev)
    per_tile.index_add_(0, tile_of, b4.long())
    q8 = torch.zeros(nt, 4, dtype=torch.long, device=dev)
    q8.index_add_(0, tile_of, q.long())
    out[name] = {
        "live_pairs": L,
        "by_block_shape": rows,
        "entries_touching_0_1_2_3_4_quadrants": [int((nq == k).sum().item()) for k in range(5)],
        "live_lanes_per_evaluated_quadrant_hist_1-8_9-16_..._57-64": hist8,
        "walk_steps_4x4_rows": {
            # a wave with 4 independent 16-lane rows, each walking 4x4 blocks: steps >= max(longest block list,
            # total / 4) per tile (perfect dynamic balance) vs. static (row r owns the 4 blocks of quadrant r)
            "sum_block_lists": int(per_tile.sum().item()),
            "steps_perfect_balance": int(torch.maximum(per_tile.max(1).values,
                                                       (per_tile.sum(1) + 3) // 4).sum().item()),
            "steps_static_quadrant_rows": int(per_tile.reshape(nt, 2, 2, 2, 2).permute(0, 1, 3, 2, 4)
                                              .reshape(nt, 4, 4).sum(2).max(1).values.sum().item()),
        },
        "walk_steps_8x8_wave": int(q8.sum().item()),
        "independent_workers": workers(live, b4),
    }
# ---- what the forward block walk actually visits: the conservative per-block masks of block_reach_mask16 (linear
# ---- lower bound of sigma over the block + margin), cut at the block's last live entry, against the exact live blocks
def lin_masks():
    thr = torch.log(255.0 * op)
    thr = thr + 0.05 + 0.02 * thr
    ok_op = (op * 255.0 >= 1.0)
    m = torch.zeros(I, 16, dtype=torch.bool, device=dev)
    for by in range(4):
        for bx in range(4):
            cx = (tx * 16).float() + 2.0 + 4 * bx
            cy = (ty * 16).float() + 2.0 + 4 * by
            dx, dy = m2[:, 0] - cx, m2[:, 1] - cy
            gx = con[:, 0] * dx + con[:, 1] * dy
            gy = con[:, 1] * dx + con[:, 2] * dy
            lb = 0.5 * (dx * gx + dy * gy) - 1.5 * (gx.abs() + gy.abs())
            m[:, 4 * by + bx] = (lb <= thr) & ok_op
    return m


ml = lin_masks()
b4f = block_any(live_fwd, 4, 4).reshape(-1, 16)
pos_in_tile = torch.arange(I, device=dev) - offs[tile_of]
last_live = torch.full((nt, 16), -1, dtype=torch.long, device=dev)
idx = pos_in_tile[:, None].expand(-1, 16).clone()
idx[~b4f] = -1
last_live.scatter_reduce_(0, tile_of[:, None].expand(-1, 16), idx, reduce="amax", include_self=True)
visited = ml & (pos_in_tile[:, None] <= last_live[tile_of])
out["forward_block_walk"] = {"exact_live_pairs": int(b4f.sum().item()), "linear_bound_pairs": int(ml.sum().item()),
                             "visited_pairs(bound, cut at the block's last live entry)": int(visited.sum().item()),
                             "workers": workers(live_fwd, visited)["16x(4x4)"]}
print(json.dumps(out, indent=1))
if args.out:
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
