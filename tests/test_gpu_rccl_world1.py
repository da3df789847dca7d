"""The collectives of the N > 1 path on the real backend (a one-rank `nccl` = RCCL group, MOBGS_FORCE_COLLECTIVES=1):
scripts/rccl_world1_check.py in its own process (it initialises a process group)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_rccl_one_rank_group_reproduces_the_collective_free_step(hip_device):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_world1_check.py")], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "RCCL world-1 check: OK" in out.stdout and "prediction identical: True" in out.stdout
