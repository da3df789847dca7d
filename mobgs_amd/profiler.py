"""Opt-in HIP-event timing of named regions on the current stream (used by bench.py for the roofline line).

Disabled by default: `region()` is then a no-op context manager, so the product path pays nothing.
Events are recorded on torch's current stream, which is the stream every mobgs kernel is launched on.
"""
from __future__ import annotations

import contextlib
from collections import defaultdict
from typing import Dict, List

import torch

_enabled = False
_events: Dict[str, List] = defaultdict(list)
counters: Dict[str, int] = {}


def enable(flag: bool = True) -> None:
    global _enabled
    _enabled = flag
    if flag:
        _events.clear()


@contextlib.contextmanager
def region(name: str):
    if not _enabled:
        yield
        return
    a = torch.cuda.Event(enable_timing=True)
    b = torch.cuda.Event(enable_timing=True)
    a.record()
    yield
    b.record()
    _events[name].append((a, b))


def summary() -> Dict[str, Dict[str, float]]:
    """{name: {calls, total_ms, avg_ms}} -- synchronises the device."""
    torch.cuda.synchronize()
    out = {}
    for name, evs in _events.items():
        tot = sum(a.elapsed_time(b) for a, b in evs)
        out[name] = {"calls": len(evs), "total_ms": tot, "avg_ms": tot / max(1, len(evs))}
    return out
