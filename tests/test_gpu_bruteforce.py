"""The HIP path against the list-free float64 restatement (oracle/gsplat_bruteforce.py): default
lists (reach culling on) and gsplat's exact bounding-box lists (culling off), the quadrant backward and both matrix-pipe
arms.  Bounds: tests/test_bruteforce_oracle_cpu.py::check_against_bruteforce."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bruteforce_reference():
    """The float64 list-free evaluation of the test scene: ~45 s of CPU work -- loaded from tests/golden/bruteforce_ref.npz,
    the oracle's own output, which the CPU suite re-derives and compares (test_stored_reference_is_the_oracles_output).
    (It was recomputed per arm: 220 of the GPU suite's 430 s -- VERDICT r4 item 4; then once per module: 45-52 s.)"""
    from test_bruteforce_oracle_cpu import _scene, load_reference
    return _scene(), load_reference()


@pytest.mark.parametrize("culling", [True, False])
@pytest.mark.parametrize("bwd_mfma", [0, 1, 2])
def test_hip_path_agrees_with_the_list_free_restatement(hip_device, bruteforce_reference, culling, bwd_mfma):
    from mobgs_amd import rendering
    from mobgs_amd.rendering import rasterization
    from test_bruteforce_oracle_cpu import NAMES, check_against_bruteforce
    scene, ref = bruteforce_reference
    s, bg, v_img, v_a, w, h = scene
    old = rendering.tuning.bwd_mfma
    rendering.set_tile_culling(culling)
    rendering.tuning.bwd_mfma = bwd_mfma
    try:
        t = {k: v.clone().to(hip_device).requires_grad_(k in NAMES) for k, v in s.items()}
        img, a, meta = rasterization(t["means"], t["quats"], t["scales"], t["opacities"], t["colors"], t["viewmats"],
                                     t["Ks"], w, h, backgrounds=bg.to(hip_device), render_mode="RGB+ED", packed=False)
        torch.autograd.backward([img, a], [v_img.to(hip_device), v_a.to(hip_device)])
    finally:
        rendering.set_tile_culling(True)
        rendering.tuning.bwd_mfma = old
    grads = {k: t[k].grad.detach().cpu() for k in NAMES}
    check_against_bruteforce(img.detach().cpu(), a.detach().cpu(), meta["radii"].cpu().numpy(), meta["means2d"].cpu(),
                             meta["conics"].cpu(), grads, scene, ref,
                             f"HIP (culling {'on' if culling else 'off'}, bwd_mfma {bwd_mfma})")
