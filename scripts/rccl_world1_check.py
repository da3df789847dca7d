"""Every collective of the N > 1 path on the REAL backend, on a one-GPU box: a ONE-rank `nccl` (= RCCL) process group and
MOBGS_FORCE_COLLECTIVES=1 make SubframeShard issue its exchanges although there is nobody to exchange with -- the sums
are identities, so the deblur step must reproduce the collective-free single-process step bit for bit: per-view
asynchronous image all-reduces on RCCL's stream (awaited before use), the autograd exchange nodes, the in-place flat
gradient all-reduce with the densification statistics, the fp32 reduction of a half-precision slice.
(Two ranks cannot share one GPU under RCCL; bench.py's N > 1 launch on one GPU uses gloo instead.)
    python scripts/rccl_world1_check.py          (GPU box; exits non-zero on a mismatch)"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402
from mobgs_amd.distributed import FlatGradients, SubframeShard  # noqa: E402


def step_result(dev, shard, seed=3):
    scam, _, stat, dyn, _ = B.build_scene(dev, 5000, 2500, 320, 240, seed=seed)
    wl = B.DeblurWorkload(dev, stat, dyn, scam, 320, 240, shard, batched=True)
    for _ in range(3):
        pred = wl.step()
    torch.cuda.synchronize()
    if isinstance(pred, (list, tuple)):   # the collective path: per-view predictions, one gradient message per view
        pred = torch.stack(list(pred))
        first = wl.view_buckets[0]        # backward_by_view leaves the sum over views (and ranks) in the first bucket
        grads = [first.flat[:first.param_floats].clone()] + ([first.flat_half.clone()] if first.flat_half.numel() else [])
    else:
        b = wl.bucket
        grads = [b.flat[:b.param_floats].clone()] + ([b.flat_half.clone()] if b.flat_half.numel() else [])
    return pred.detach().clone(), grads


def overlap_check(dev, shard):
    """The per-view image exchange is enqueued from the communication stream, not from the compute stream, and the
    exchange of view 0 completes WHILE view 1 is still being produced: device timestamps of the
    exchange's start / done events (MOBGS_COMM_LOG=1) against two events around the units of view 1."""
    from mobgs_amd import distributed as D
    os.environ["MOBGS_COMM_LOG"] = "1"
    D.comm_log.clear()
    a = torch.randn(2048, 2048, device=dev)
    marks = {}

    def unit(v, k):
        if v == 1 and "begin" not in marks:
            marks["begin"] = torch.cuda.Event(enable_timing=True)
            marks["begin"].record()
        x = a
        for _ in range(6):          # a few hundred microseconds of compute per unit on the current stream
            x = (x @ a) * 1e-3
        img = x.reshape(-1)[:3 * 240 * 320].reshape(3, 240, 320).clone().requires_grad_(True) * 1.0
        if v == 1:
            marks["end"] = torch.cuda.Event(enable_timing=True)
            marks["end"].record()
        return img

    shard.render_blurry_views(unit, 2, 9, like=torch.zeros(3, 240, 320, device=dev), overlap=True)
    torch.cuda.synchronize()
    os.environ.pop("MOBGS_COMM_LOG", None)
    images = [(s, d) for tag, s, d in D.comm_log if tag == "image"]
    side, main = D.comm_stream(dev), torch.cuda.current_stream(dev)
    view1_ms = marks["begin"].elapsed_time(marks["end"])
    done0_ms = marks["begin"].elapsed_time(images[0][1])     # exchange of view 0 done, relative to the start of view 1
    ok = len(images) == 2 and side != main and side.cuda_stream != main.cuda_stream and done0_ms < view1_ms
    print(f"overlap: {len(images)} image exchanges on stream {side.cuda_stream:#x} (compute stream {main.cuda_stream:#x}); "
          f"view 1 took {view1_ms:.3f} ms on the compute stream, the exchange of view 0 was complete {done0_ms:.3f} ms after "
          f"view 1 began -> {'inside' if done0_ms < view1_ms else 'AFTER'} it")
    return ok


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    os.environ.pop("MOBGS_FORCE_COLLECTIVES", None)
    ref_pred, ref_grads = step_result(dev, SubframeShard(world_size=1, rank=0))
    import socket
    with socket.socket() as sk:   # a free port
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    os.environ["MOBGS_FORCE_COLLECTIVES"] = "1"
    shard = SubframeShard()
    assert shard.collective and shard.world == 1 and dist.get_backend() == "nccl"
    pred, grads = step_result(dev, shard)
    ok = torch.equal(pred, ref_pred)
    print("prediction identical:", ok, "max diff", float((pred - ref_pred).abs().max()))
    for a, b in zip(grads, ref_grads):
        same = torch.equal(a, b)
        # (the weighted plan of the collective path walks the units in another order than the plain loop: the flat
        # buffer is the same sum in a different order)
        close = torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
        print(f"flat gradient buffer {tuple(a.shape)} {a.dtype}: identical {same}, close {close}, "
              f"max diff {float((a.float() - b.float()).abs().max()):.3e} of {float(b.float().abs().max()):.3e}")
        ok = ok and close
    # the half-precision slice (reduced in fp32) and a plain parameter list
    p16 = torch.randn(1000, device=dev).half().requires_grad_(True)
    p32 = torch.randn(77, device=dev).requires_grad_(True)
    bucket = FlatGradients([p32, p16])
    bucket.zero()
    p32.grad.copy_(torch.arange(77, device=dev, dtype=torch.float32))
    p16.grad.copy_(torch.linspace(-3, 3, 1000, device=dev).half())
    want32, want16 = p32.grad.clone(), p16.grad.clone()
    works = shard.all_reduce_gradients(bucket, async_op=True)
    for w in works or []:
        w.wait()
    torch.cuda.synchronize()
    ok = ok and torch.equal(p32.grad, want32) and torch.equal(p16.grad, want16)
    q = torch.randn(50, device=dev, requires_grad=True)
    q.grad = torch.ones_like(q)
    shard.all_reduce_gradients([q])
    ok = ok and torch.equal(q.grad, torch.ones_like(q))
    ok = overlap_check(dev, shard) and ok
    print("RCCL world-1 check:", "OK" if ok else "MISMATCH")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
