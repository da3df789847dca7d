import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def hip_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")


class _Selection:
    """What a `kernel_selection` test gets: the name of the arm and, after the body ran, check() -- which ASSERTS from
    rendering.path_log (mobgs_raster_path: the launchers' own decision functions + the tile schedule read back) that the
    10- / 12-channel compositing passes took the kernels the arm names."""

    def __init__(self, name):
        self.name = name

    def passes(self):
        from mobgs_amd import rendering
        return [e for e in (rendering.path_log or []) if e["D"] in (10, 12)]

    def check(self, need_bwd=True):
        ps = self.passes()
        assert any(e["dir"] == "fwd" for e in ps), "no 10 / 12-channel forward pass was logged"
        if need_bwd:
            assert any(e["dir"] == "bwd" for e in ps), "no 10 / 12-channel backward pass was logged"
        for e in ps:
            if self.name == "headline":
                # the benchmark's selection: one wave per tile (no heavy tiles), block-walk forward (with the decoder
                # epilogue where render() fuses it), quadrant backward raster_bwd_kernel
                assert e["heavy_tiles"] == 0 and e["heavy_len"] == 0, e
                assert e["fwd_kernel"] == "blocks" and e["bwd_kernel"] == "quadrant", e
            else:
                # small grids (every fixture is one): all tiles heavy, matrix-pipe backward where it exists
                assert e["n_tiles"] > 1024 or e["heavy_len"] == 1, e
        return ps


@pytest.fixture(params=["default", "headline"])
def kernel_selection(request):
    """Runs the test once under the library's default kernel selection -- on the fixtures' small grids (<= 1024 tiles)
    that is four waves per tile + the matrix-pipe backward -- and once under the selection the BENCHMARK runs at
    1352x1014: MobgsTuning.heavy_tile_len = 0 (one wave per tile: raster_fwd_blocks with the decoder
    epilogue) and bwd_mfma = 0 (the quadrant kernel raster_bwd_kernel)."""
    from mobgs_amd import rendering
    saved = (rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma, rendering.path_log)
    if request.param == "headline":
        rendering.tuning.heavy_tile_len = 0
        rendering.tuning.bwd_mfma = 0
    rendering.path_log = []
    try:
        yield _Selection(request.param)
    finally:
        rendering.tuning.heavy_tile_len, rendering.tuning.bwd_mfma, rendering.path_log = saved
