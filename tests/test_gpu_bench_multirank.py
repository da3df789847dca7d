"""The N > 1 code path of bench.py -- the driver's scaling run: `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N ...` -- exercised on ONE GPU: all ranks share the device and exchange through gloo
(MOBGS_BENCH_SHARE_GPU=1 MOBGS_BENCH_BACKEND=gloo; the line labels itself a functional check).  3 and 8 ranks: with 18
(view, sub-frame) units some ranks own a single latent sub-frame of a view (a batch of one), some none of a view."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [3, 8])
def test_bench_sharded_path_runs_and_reports(hip_device, world):
    env = dict(os.environ, MOBGS_BENCH_SHARE_GPU="1", MOBGS_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    import socket
    with socket.socket() as sk:   # a free port (a fixed one may sit in TIME_WAIT after an earlier run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--prewarm", "2",
           "--ns", "6000", "--nd", "3000", "--width", "320", "--height", "240"]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == world and line["steps"] == 2 and line["scaling"] == "strong"
    assert line["value"] > 0 and line["config"]["renders_per_step"] == 18
    assert "FUNCTIONAL CHECK" in line["data"]
    # the anchor of the scaling curve is measured in the same run: the N > 1 workload with every unit on one GPU
    assert line["scale_anchor"] > 0 and "scaling_efficiency_vs_anchor" in line and "scale_anchor_field" in line
    # ... and the line explains itself: per-rank busy / exposed-exchange times, the partition's own ceiling, and the
    # in-run check that every rank holds the same all-reduced prediction bit for bit
    diag = line["scaling_diagnostics"]
    assert len(diag["per_rank"]) == world and all(r["busy_ms"] > 0 for r in diag["per_rank"])
    assert {"image_exchange_exposed_ms", "grad_exchange_exposed_ms", "iteration_ms"} <= set(diag["per_rank"][0])
    assert 0.5 < diag["planned_efficiency_ceiling"] <= 1.0 and len(diag["planned_loads_latent_render_units"]) == world
    assert diag["prediction_bit_identical_across_ranks"] is True
