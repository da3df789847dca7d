"""A/B of the two backward compositors on one box: gradients of the same forward with MobgsTuning.bwd_mfma = 0 / 1
(difference relative to the largest magnitude of each gradient tensor), then the step time of both arms.
usage: python scripts/check_bwd_mfma.py [N=300000] [width height]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mobgs_amd.synth import SynthCamera, splat_inputs
from mobgs_amd import rendering
from mobgs_amd.rendering import rasterization

dev = torch.device("cuda")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
cam = SynthCamera().scaled(int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else SynthCamera()
s = {k: v.to(dev) for k, v in splat_inputs(N, cam, 0, 9).items()}
names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
for k in names:
    s[k].requires_grad_(True)
bg = torch.rand(1, 9, device=dev)
g = torch.Generator().manual_seed(100)
v_img = torch.randn(1, cam.height, cam.width, 10, generator=g).to(dev)
v_a = torch.randn(1, cam.height, cam.width, 1, generator=g).to(dev)


def step(with_alpha=True):
    for k in names:
        s[k].grad = None
    img, a, meta = rasterization(s["means"], s["quats"], s["scales"], s["opacities"], s["colors"], s["viewmats"], s["Ks"],
                                 cam.width, cam.height, packed=False, backgrounds=bg, render_mode="RGB+ED")
    loss = (img * v_img).sum()
    if with_alpha:
        loss = loss + (a * v_a).sum()
    loss.backward()
    return meta


res = {}
ARMS = [int(a) for a in os.environ.get("MOBGS_ARMS", "0,1").split(",")]
for arm in ARMS:
    rendering.tuning.bwd_mfma = arm
    meta = step()
    torch.cuda.synchronize()
    res[arm] = {k: s[k].grad.detach().clone() for k in names}
print("I =", meta["flatten_ids"].numel(), "tiles", meta["tile_width"] * meta["tile_height"])
worst = 0.0
for k in names:
    for arm in ARMS[1:]:
        a, b = res[ARMS[0]][k], res[arm][k]
        d = (a - b).abs().max().item()
        m = a.abs().max().item()
        rel = ((a - b).abs() / (a.abs() + 1e-4 * m)).max().item()
        worst = max(worst, d / m)
        print(f"{k:10s} arm {arm}: max|d| {d:.3e}  max|ref| {m:.3e}  d/max {d/m:.2e}  worst elementwise rel (floor 1e-4 max) {rel:.2e}  nan {int(torch.isnan(b).sum())}")
print("WORST d/max", f"{worst:.2e}")
for arm in ARMS + ARMS:
    rendering.tuning.bwd_mfma = arm
    for _ in range(5):
        step(False)
    torch.cuda.synchronize()
    t0 = time.time()
    K = 30
    for _ in range(K):
        step(False)
    torch.cuda.synchronize()
    print(f"bwd_mfma={arm}: fwd+bwd {(time.time() - t0) / K * 1e3:.3f} ms")

# development builds (-DMOBGS_MFMA_TIMING) keep per-phase cycle sums
import ctypes
lib = ctypes.CDLL(os.environ.get("MOBGS_LIB", "mobgs_amd/csrc/libmobgs_hip.so"))
if hasattr(lib, "mobgs_debug_mfma_timing"):
    buf = (ctypes.c_ulonglong * 8)()
    rendering.tuning.bwd_mfma = ARMS[-1]
    step(False); torch.cuda.synchronize()
    lib.mobgs_debug_mfma_timing(buf, 1)
    step(False); torch.cuda.synchronize()
    lib.mobgs_debug_mfma_timing(buf, 0)
    t = list(buf)
    tot = sum(t[:5])
    names_ = ["prologue", "staging", "walk", "reduce", "flush"]
    print("timing (cycles summed over waves):", {n: f"{v / tot:.3f}" for n, v in zip(names_, t[:5])}, "waves", t[5], "groups", t[6], "evals", t[7],
          "cycles/wave", tot // max(t[5], 1), "walk cycles/eval", t[2] / max(t[7], 1), "reduce cycles/group", t[3] / max(t[6], 1))
