"""Run-to-run determinism of the class-restricted backward compositor at operator level (704x400, lists of ~200)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch
from mobgs_amd import rendering as R
from mobgs_amd.synth import SynthCamera, splat_inputs
dev = torch.device("cuda:0")
W, H, N, Ns = 704, 400, 45000, 30000
scam = SynthCamera().scaled(W, H)
s = splat_inputs(N, scam, 4, 9)
v = torch.randn(1, H, W, 10, generator=torch.Generator().manual_seed(5)).to(dev)
def run(class_passes):
    R.CLASS_PASSES = class_passes
    t = {k: x.to(dev).clone().requires_grad_(k in ("means", "quats", "scales", "opacities", "colors")) for k, x in s.items()}
    sp = R.SharedProjection(t["means"], t["quats"], t["scales"], t["opacities"], t["viewmats"], t["Ks"], W, H)
    imgs, alps = sp.composite_layers(t["colors"], Ns, torch.zeros(1, 9, device=dev), want_static=True, want_dynamic=False)
    (imgs[1] * v).sum().backward()
    return t
t0, t1, t2 = run(False), run(True), run(True)
for name, a, b in (("layers vs class", t0, t1), ("class vs class", t1, t2)):
    out = []
    for k in ("means", "opacities", "colors"):
        d = (a[k].grad - b[k].grad).abs().reshape(N, -1).max(1).values
        sc = float(a[k].grad.abs().max())
        out.append(f"{k}: {int((d > 1e-4 * sc).sum())} bad, max {float(d.max()):.3g} / {sc:.3g}")
    print(os.environ.get("MOBGS_LIB", "main")[-24:], name, " | ".join(out))
