"""Randomised regimes the fixed-seed parity tests do not visit (scripts/soak_parity.py): tiny / huge / thin splats,
opacities at the 1/255 and 0.999 edges, splats at the near plane, equal depths, rotated cameras, ragged image sizes,
1..11 channels, every render mode, with and without background -- against the C oracle.  radii and per-tile lists
must be bit-equal; images and gradients within the flip-aware tolerances."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [0, 7])
def test_random_regimes_against_the_c_oracle(hip_device, seed):
    import soak_parity
    failed, msgs = soak_parity.soak(40, seed, hip_device, verbose=False)
    assert failed == 0, "\n".join(msgs)


@pytest.mark.parametrize("arm,heavy", [(1, 0), (2, None)])
def test_random_regimes_with_the_matrix_pipe_backward(hip_device, arm, heavy):
    """The same soak with MobgsTuning.bwd_mfma forced (round 4): arm 1 without heavy tiles = the wave-per-tile kernel on
    every grid, arm 2 = the four-wave team on every tile (the default policy -- team on grids of <= 1024 tiles -- is what
    the unparametrised soak above runs, since most of its images are that small)."""
    import soak_parity
    from mobgs_amd import rendering
    old = (rendering.tuning.bwd_mfma, rendering.tuning.heavy_tile_len)
    rendering.tuning.bwd_mfma = arm
    if heavy is not None:
        rendering.tuning.heavy_tile_len = heavy
    try:
        failed, msgs = soak_parity.soak(24, 11 + arm, hip_device, verbose=False)
    finally:
        rendering.tuning.bwd_mfma, rendering.tuning.heavy_tile_len = old
    assert failed == 0, "\n".join(msgs)


def test_random_render_calls_against_the_torch_restatement(hip_device):
    """render() (boundary B1) in random regimes (scripts/soak_render.py): camera times on and between the spline's
    knots, exposure offsets pushing the time outside [0, 1], 4..12 control points, lean and train mode, random
    backgrounds and cameras -- against oracle/render_torch.py."""
    import soak_render
    failed, msgs = soak_render.soak(4, 3, hip_device, verbose=False)
    assert failed == 0, "\n".join(msgs)


@pytest.mark.parametrize("which", ["loss", "normals", "deform", "blce"])
def test_random_side_kernel_regimes(hip_device, which):
    """scripts/soak_misc.py: fused L1 + SSIM on images from 1x1 up (smaller than the 11x11 window, ragged against the
    kernel's tiles, batches, constant images), normals from depth on 3x3.. images with skewed intrinsics,
    deform_network with point counts around its 64-point tiles, points outside the bounding box and times 0 / 1, the
    fused BLCE kernels with random parameters / view counts / poses -- against the CPU restatements (BLCE: the PyTorch
    module)."""
    import soak_misc
    failed, msgs = soak_misc.soak(which, 16, 1, hip_device, verbose=False)
    assert failed == 0, "\n".join(msgs)


def test_random_get_flow_calls_against_the_torch_restatement(hip_device):
    """get_flow() and get_flow_many() in random regimes (scripts/soak_render.py --flow): all four outputs and the
    leaf gradients against oracle/render_torch.get_flow."""
    import soak_render
    # (2 cases here, 12-18 s each of CPU oracle time; the long form is `python scripts/soak_render.py --flow --cases 30`)
    failed, msgs = soak_render.soak(2, 5, hip_device, verbose=False, flow=True)
    assert failed == 0, "\n".join(msgs)


def test_random_full_size_scenes_with_long_lists(hip_device):
    """scripts/soak_parity.py --large: 60 k .. 300 k splats at 512x288 .. 1352x1014, enlarged and clustered so that
    per-tile lists reach thousands to tens of thousands of entries (every sort build, heavy tiles): lists bit-equal
    to the C oracle's, images and gradients within tolerance."""
    import soak_parity
    failed, msgs = soak_parity.soak(4, 2, hip_device, verbose=False, large=True)
    assert failed == 0, "\n".join(msgs)


@pytest.mark.parametrize("w,h,C", [(2064, 1040, 1), (1100, 720, 3)])
def test_grids_beyond_8192_tiles_match_the_c_oracle(hip_device, w, h, C):
    """129 x 65 = 8385 tiles (and 3 cameras x 69 x 45 = 9315): more tiles than the LDS-ranked binning and the schedule
    workgroup's LDS table hold, i.e. the direct-atomics form of bin_kernel over the eight counter copies and the
    re-reading form of the tile order.  Lists bit-equal to the C oracle (tile culling off), image / gradients within
    the soak's tolerances."""
    import math

    import numpy as np
    import soak_parity
    from mobgs_amd.synth import SynthCamera, splat_inputs
    rng = np.random.default_rng(5)
    cam = SynthCamera().scaled(w, h)
    s = splat_inputs(7000, cam, 11, 3)
    if C > 1:
        vms = []
        for i in range(C):
            ang = 0.1 * (i - 1)
            vm = torch.eye(4)
            vm[:3, :3] = torch.tensor([[math.cos(ang), 0, math.sin(ang)], [0, 1, 0], [-math.sin(ang), 0, math.cos(ang)]])
            vm[:3, 3] = torch.tensor([0.05 * i, -0.02 * i, 0.1 * i])
            vms.append(vm)
        s["viewmats"] = torch.stack(vms)
        s["Ks"] = s["Ks"].expand(C, 3, 3).contiguous()
    case = dict(w=w, h=h, n=7000, channels=3, mode="RGB+ED", regime="plain", s=s,
                bg=torch.rand(C, 3, generator=torch.Generator().manual_seed(3)), X=4, idx=int(rng.integers(1000)), C=C)
    problems, _ = soak_parity.run_case(case, hip_device)
    assert not problems, problems
