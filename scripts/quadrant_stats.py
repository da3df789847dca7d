"""How much of the compositor's pair evaluation is spent on quadrants / pixels a splat cannot touch?
For every (tile, splat) entry of the benchmark lists: which of the four 8x8 quadrants hold at least one pixel with
alpha >= 1/255 (ignoring transmittance), and the fraction of passing pixels.  Sizes the gain of a per-quadrant
reach mask in raster_fwd / raster_bwd."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.synth import SynthCamera, splat_inputs
from mobgs_amd.rendering import rasterization

dev = torch.device("cuda")
cam = SynthCamera()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
s = {k: v.to(dev) for k, v in splat_inputs(N, cam, 0, 9).items()}
bg = torch.zeros(1, 9, device=dev)
img, a, meta = rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmats"], s["Ks"],
                             cam.width, cam.height, packed=False, backgrounds=bg, render_mode="RGB+ED")
ids = meta["flatten_ids"].long()
offs = meta["isect_offsets"].reshape(-1).long()
tw = meta["tile_width"]
nt = offs.numel()
I = ids.numel()
ends = torch.cat([offs[1:], torch.tensor([I], device=dev)])
tile_of = torch.repeat_interleave(torch.arange(nt, device=dev), ends - offs)
m2 = meta["means2d"].reshape(-1, 2)[ids]
con = meta["conics"].reshape(-1, 3)[ids]
op = meta["opacities"].reshape(-1)[ids] if meta["opacities"].numel() > 1 else s["opacities"][ids]
ty, tx = tile_of // tw, tile_of % tw
lx = torch.arange(16, device=dev) + 0.5
quad_any = torch.zeros(I, 4, dtype=torch.bool, device=dev)
npass = torch.zeros(I, device=dev)
for c0 in range(0, I, 200000):
    sl = slice(c0, min(I, c0 + 200000))
    px = (tx[sl] * 16)[:, None, None] + lx[None, None, :]
    py = (ty[sl] * 16)[:, None, None] + lx[None, :, None]
    dx = m2[sl, 0][:, None, None] - px
    dy = m2[sl, 1][:, None, None] - py
    sig = 0.5 * (con[sl, 0][:, None, None] * dx * dx + con[sl, 2][:, None, None] * dy * dy) + con[sl, 1][:, None, None] * dx * dy
    al = torch.clamp(op[sl][:, None, None] * torch.exp(-sig), max=0.999)
    ok = (sig >= 0) & (al >= 1 / 255) & (px < cam.width) & (py < cam.height)
    npass[sl] = ok.float().sum((1, 2))
    q = ok.reshape(-1, 2, 8, 2, 8).any(dim=4).any(dim=2)  # [n, qy, qx]
    quad_any[sl] = q.reshape(-1, 4)
print("entries", I, "tiles", nt)
print("fraction of quadrants touched      : %.3f" % quad_any.float().mean().item())
print("fraction of pixels passing         : %.3f" % (npass.sum().item() / (I * 256)))
print("entries touching 0/1/2/3/4 quadrants:", [round((quad_any.sum(1) == k).float().mean().item(), 3) for k in range(5)])
print("pass fraction inside touched quadrants: %.3f" % (npass.sum().item() / (quad_any.sum().item() * 64)))
