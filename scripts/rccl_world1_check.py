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
    grads = [b.clone() for b in wl.bucket.buffers()]
    return pred.detach().clone(), grads


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
    print("RCCL world-1 check:", "OK" if ok else "MISMATCH")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
