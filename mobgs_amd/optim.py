"""One Adam step for several torch.optim.Adam optimisers in ONE kernel launch (include/mobgs_hip.h K16).

The reference steps three optimisers per iteration (/root/reference/train.py:790-807); with one-tensor parameter groups
(scene/gaussian_model.py:598-617) torch's multi-tensor path degenerates to ~8 launches per group: 3.3 ms per iteration at
300 k Gaussians.  `fused_adam_step` performs exactly torch.optim.Adam's update (amsgrad = False, weight_decay = 0,
maximize = False) for every fp32 CUDA parameter of the given optimisers, on their own state tensors (`exp_avg`,
`exp_avg_sq`, `step` -- created on first use like torch does, so state_dict() / densification surgery see what they expect),
and falls back to `optimizer.step()` for anything else (other dtypes, weight decay, amsgrad, CPU tensors).
"""
from __future__ import annotations

import ctypes
import math
from typing import Iterable, List

import torch

from . import _lib
from ._lib import check, stream


class _AdamTensor(ctypes.Structure):
    _fields_ = [("param", ctypes.c_void_p), ("grad", ctypes.c_void_p), ("exp_avg", ctypes.c_void_p),
                ("exp_avg_sq", ctypes.c_void_p), ("n", ctypes.c_int64), ("step_size", ctypes.c_float),
                ("bias2_sqrt", ctypes.c_float)]


def _fusable(opt: torch.optim.Optimizer, group: dict, p: torch.Tensor) -> bool:
    return (isinstance(opt, torch.optim.Adam) and not group.get("amsgrad", False) and not group.get("maximize", False)
            and group.get("weight_decay", 0) == 0 and not group.get("capturable", False)
            and not group.get("differentiable", False) and p.is_cuda and p.dtype == torch.float32
            and p.is_contiguous() and p.grad is not None and p.grad.dtype == torch.float32
            and p.grad.is_contiguous() and not p.grad.is_sparse)


@torch.no_grad()
def fused_adam_step(optimizers: Iterable[torch.optim.Optimizer]) -> int:
    """Step all the optimisers; -> number of tensors that went through the fused kernel."""
    lib = _lib.load()
    batches = {}   # (beta1, beta2, eps) -> descriptors
    keep: List[torch.Tensor] = []
    fused = 0
    for opt in optimizers:
        leftovers = []
        for group in opt.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not _fusable(opt, group, p):
                    leftovers.append(p)
                    continue
                st = opt.state[p]
                if len(st) == 0:   # torch.optim.Adam._init_group
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                t = float(st["step"])
                lr = float(group["lr"])
                step_size = lr / (1.0 - b1 ** t)
                bias2_sqrt = math.sqrt(1.0 - b2 ** t)
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    st["step"] -= 1
                    leftovers.append(p)
                    continue
                fused += 1
                if p.numel() == 0:   # (an empty group, e.g. f_rest of a degree-0 model: state and step count only)
                    continue
                keep += [p, p.grad, m, v]
                batches.setdefault((float(b1), float(b2), float(group["eps"])), []).append(
                    _AdamTensor(p.data_ptr(), p.grad.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), step_size,
                                bias2_sqrt))
        if leftovers:  # torch's own step for what the kernel does not cover (grads of the fused ones hidden meanwhile)
            hidden = [(p, p.grad) for g in opt.param_groups for p in g["params"]
                      if p.grad is not None and all(p is not q for q in leftovers)]
            for p, _ in hidden:
                p.grad = None
            try:
                opt.step()
            finally:
                for p, g in hidden:
                    p.grad = g
    if batches:
        # the kernel writes the parameters through raw pointers: Tensor._version does not move, which is what render-side
        # caches key on (gaussian_renderer._shared_mid_state: ADVICE r5) -- tell them
        from . import gaussian_renderer as _GR
        _GR.parameters_changed()
    for (b1, b2, eps), descs in batches.items():
        for i in range(0, len(descs), 64):
            chunk = descs[i:i + 64]
            arr = (_AdamTensor * len(chunk))(*chunk)
            check(lib.mobgs_adam_step(len(chunk), arr, b1, b2, eps, stream()), "mobgs_adam_step")
    return fused


class FusedAdam(torch.optim.Adam):
    """torch.optim.Adam whose step() is the one-launch kernel above: what the modules of this package that stand in for
    the reference's own (densify.TrainableGaussians for scene/gaussian_model.py GaussianModel.training_setup, :598-617;
    blce.blceKernel for scene/blce.py) hand to an UNCHANGED training loop -- /root/reference/train.py:790-807 calls
    `optimizer.step()` on whatever training_setup() created.  Same update, same state tensors
    (`exp_avg`, `exp_avg_sq`, `step`), state_dict()-compatible; anything the kernel does not cover (closure, amsgrad, weight
    decay, non-fp32 / CPU tensors) goes through torch's own step.  MOBGS_FUSED_ADAM=0: plain torch.optim.Adam.step()."""

    @torch.no_grad()
    def step(self, closure=None):
        import os
        if closure is not None or os.environ.get("MOBGS_FUSED_ADAM", "1") == "0":
            return super().step(closure)
        # (fused_adam_step falls back to torch's step for the tensors it cannot take: reach it through the base class,
        # not through this override)
        real, self.step = self.step, super().step
        try:
            fused_adam_step([self])
        finally:
            self.step = real
        return None
