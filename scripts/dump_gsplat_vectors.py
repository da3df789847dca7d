#!/usr/bin/env python
"""Close the one open parity pin: dump input/output vectors from a REAL `gsplat==1.4.0` install (the version the
reference pins, /root/reference/README.md:26) in the fixture format of tests/golden.

The build container and the GPU boxes of this project have no gsplat (pip dependency, not vendored, no network), so
oracle/gsplat_torch.py and oracle/gsplat_cpu.c -- the restatements every rasterizer parity test compares against --
are pinned only to each other and to analytic known answers.  On any machine that has gsplat 1.4.0 and a CUDA/HIP
device:

    pip install gsplat==1.4.0
    python scripts/dump_gsplat_vectors.py            # writes tests/golden/gsplat/case_*.npz

The test suite picks the files up automatically: tests/test_gsplat_vectors.py compares the torch oracle (CPU run)
and the HIP path (`-m gpu` run) against them; without the files those tests are skipped and the pin stays open.
Each case stores: inputs (means, quats, scales, opacities, colors, viewmats, Ks, size, mode, background), the
cotangents, outputs (render, alphas, radii, means2d, depths, conics, tiles_per_gauss, isect_ids, flatten_ids,
isect_offsets) and the gradients of all differentiable inputs incl. viewmats and means2d.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mobgs_amd.synth import SynthCamera, splat_inputs  # noqa: E402

CASES = [  # name, n, W, H, channels, render_mode, background?, seed, camera offset
    ("rgb_ed_9ch", 1200, 104, 72, 9, "RGB+ED", True, 4, True),
    ("rgb_1ch", 1200, 104, 72, 1, "RGB", True, 5, True),
    ("rgb_2ch_nobg", 800, 96, 64, 2, "RGB", False, 6, False),
    ("rgb_d_3ch", 1000, 128, 80, 3, "RGB+D", True, 7, True),
    ("ed_only", 600, 64, 48, 3, "ED", False, 8, False),
    ("dense_tile", 3000, 64, 64, 3, "RGB", True, 9, False),
]


def main():
    import gsplat
    from gsplat.rendering import rasterization
    assert gsplat.__version__.startswith("1.4"), f"need gsplat 1.4.x (the reference's pin), found {gsplat.__version__}"
    dev = torch.device("cuda")
    out_dir = os.path.join(ROOT, "tests", "golden", "gsplat")
    os.makedirs(out_dir, exist_ok=True)
    names = ["means", "quats", "scales", "opacities", "colors", "viewmats"]
    for name, n, W, H, ch, mode, use_bg, seed, offset in CASES:
        s = splat_inputs(n, SynthCamera().scaled(W, H), seed, ch)
        if name == "dense_tile":  # > 256 splats through single tiles: batching and the 1e-4 stop
            s["means"][:, :2] *= 0.15
        s["viewmats"] = s["viewmats"].clone()
        if offset:
            s["viewmats"][0, :3, 3] = torch.tensor([0.02, -0.01, 0.05])
        g = torch.Generator().manual_seed(seed + 1)
        bg = torch.rand(1, ch, generator=g) if use_bg else None
        t = {k: v.to(dev).clone().requires_grad_(k in names) for k, v in s.items()}
        img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                     t["Ks"], W, H, packed=False, render_mode=mode,
                                     backgrounds=None if bg is None else bg.to(dev))
        meta["means2d"].retain_grad()
        v_img = torch.randn(img.shape, generator=g)
        v_a = torch.randn(a.shape, generator=g)
        ((img * v_img.to(dev)).sum() + (a * v_a.to(dev)).sum()).backward()
        arrays = {"in_" + k: v.numpy() for k, v in s.items()}
        arrays.update({"in_size": np.array([W, H]), "in_mode": np.array(mode), "cot_render": v_img.numpy(),
                       "cot_alphas": v_a.numpy(), "out_render": img.detach().cpu().numpy(),
                       "out_alphas": a.detach().cpu().numpy(), "gsplat_version": np.array(gsplat.__version__)})
        if bg is not None:
            arrays["in_backgrounds"] = bg.numpy()
        for k in ("radii", "means2d", "depths", "conics", "tiles_per_gauss", "isect_ids", "flatten_ids",
                  "isect_offsets"):
            arrays["out_" + k] = meta[k].detach().cpu().numpy()
        for k in names:
            arrays["grad_" + k] = t[k].grad.cpu().numpy()
        arrays["grad_means2d"] = meta["means2d"].grad.cpu().numpy()
        path = os.path.join(out_dir, f"case_{name}.npz")
        np.savez_compressed(path, **arrays)
        print("wrote", path)


if __name__ == "__main__":
    main()
