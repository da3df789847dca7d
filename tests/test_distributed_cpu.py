"""World-size-2 gloo test (CPU) of the sub-frame sharding: the sharded blurry view and all parameter gradients
equal the single-process result (train.py:502-541 semantics: mean of K latent renders + 1e-10)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mobgs_amd.distributed import SubframeShard


def _toy_render(params, k):
    """A differentiable stand-in for render(warped_cam[k], delta_exposure[k])["render"]: [3,6,8]."""
    w, b = params
    t = (k - 4) / 4.0
    grid = torch.linspace(0, 1, 3 * 6 * 8).reshape(3, 6, 8)
    return torch.sigmoid(w[0] * grid + w[1] * t + b * (grid * t))


def _single_process(K):
    torch.manual_seed(0)
    w = torch.randn(2, requires_grad=True)
    b = torch.randn(1, requires_grad=True)
    pred = torch.stack([_toy_render((w, b), k) for k in range(K)]).mean(0) + 1e-10
    target = torch.full_like(pred, 0.3)
    loss = (pred - target).abs().mean()
    loss.backward()
    return pred.detach(), w.grad.clone(), b.grad.clone()


def _worker(rank, world, port, K, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        w = torch.randn(2, requires_grad=True)
        b = torch.randn(1, requires_grad=True)
        shard = SubframeShard()
        assert shard.world == world and shard.rank == rank
        like = torch.zeros(3, 6, 8)
        pred = shard.render_blurry_view(lambda k: _toy_render((w, b), k), K, like=like)
        target = torch.full_like(pred, 0.3)
        loss = (pred - target).abs().mean()
        if loss.requires_grad:  # a rank that owns no sub-frame (world > K) has nothing to back-propagate ...
            loss.backward()
        shard.all_reduce_gradients([w, b])  # ... but still takes part in the gradient all-reduce
        q.put((rank, pred.detach(), w.grad.clone(), b.grad.clone()))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("K", [9, 1])
def test_subframe_sharding_world2_matches_single_process(K):
    ref_pred, ref_w, ref_b = _single_process(K)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, K, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, pred, gw, gb in results:
        assert torch.allclose(pred, ref_pred, atol=1e-6), f"rank {rank}: blurry prediction differs"
        assert torch.allclose(gw, ref_w, atol=1e-6) and torch.allclose(gb, ref_b, atol=1e-6), f"rank {rank}: grads"


def test_unit_partition():
    s = [SubframeShard(8, r) for r in range(8)]
    units = [u for sh in s for u in sh.units(9)]
    assert sorted(units) == list(range(9))
    assert max(len(sh.units(9)) for sh in s) == 2 and s[0].units(9) == [0, 8]
    assert SubframeShard(1, 0).units(9) == list(range(9))
    one = torch.ones(3, 2, 2)
    assert torch.equal(SubframeShard(1, 0).mean_of_subframes(one, 1), one)
    assert torch.allclose(SubframeShard(1, 0).mean_of_subframes(9 * one, 9), one + 1e-10)
