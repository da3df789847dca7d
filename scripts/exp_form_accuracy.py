"""How far are two builds of the compositors from the EXACT gradients?  The scene of
tests/test_gpu_operator_parity.py::test_rasterization_backward[RGB-2-False] (1500 splats, 120 x 88, the case whose
camera gradient is the most cancellation-prone aggregate of the suite) on the HIP path with the library given by
MOBGS_LIB (or the in-tree one), against the float64 list-free oracle (oracle/gsplat_bruteforce.py) and against the fp32
C restatement of upstream's kernels (oracle/gsplat_cpu.c).  Test infrastructure: GPU box only.
    python scripts/exp_form_accuracy.py            (prints one line per gradient)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from mobgs_amd.synth import SynthCamera, splat_inputs
    from mobgs_amd.rendering import rasterization
    from oracle import gsplat_bruteforce as BF
    from oracle import gsplat_cpu as Cc
    n, w, h, channels = 1500, 120, 88, 2
    s = splat_inputs(n, SynthCamera().scaled(w, h), 4, channels)
    names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
    g = torch.Generator().manual_seed(7)
    v_img = torch.randn((1, h, w, channels), generator=g)
    v_a = torch.randn((1, h, w, 1), generator=g)

    def run(fn, dev, dtype):
        t = {k: v.to(dev).to(dtype if v.is_floating_point() else v.dtype).clone().requires_grad_(k in names) for k, v in s.items()}
        img, a, meta = fn(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"], t["Ks"], w, h,
                          packed=False, backgrounds=None, render_mode="RGB")
        ((img * v_img.to(dev).to(img.dtype)).sum() + (a * v_a.to(dev).to(a.dtype)).sum()).backward()
        return {k: t[k].grad.detach().cpu().double() for k in names}

    hip = run(rasterization, torch.device("cuda:0"), torch.float32)
    exact = run(BF.rasterization, torch.device("cpu"), torch.float64)
    r = Cc.rasterization_fwd_bwd(*(s[k].numpy() for k in ["means", "quats", "scales", "opacities", "colors", "viewmats", "Ks"]),
                                 w, h, backgrounds=None, render_mode="RGB", v_render=v_img.numpy(), v_alphas=v_a[..., 0].numpy())
    print("library:", os.environ.get("MOBGS_LIB", "(in tree)"))
    for k in names:
        c = torch.from_numpy(r["v_" + k]).double()
        m = float(exact[k].abs().max())
        print(f"{k:10s} max |g| {m:10.4e}   HIP - exact {float((hip[k] - exact[k]).abs().max()) / m:9.2e}   "
              f"C oracle - exact {float((c - exact[k]).abs().max()) / m:9.2e}   HIP - C oracle "
              f"{float((hip[k] - c).abs().max()) / m:9.2e}   (of max)")


if __name__ == "__main__":
    main()
