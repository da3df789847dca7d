"""Run the reference's OWN Python glue on CPU, in this container only (test infrastructure).

/root/reference needs CUDA + gsplat + several absent third-party packages.  This harness makes its
`gaussian_renderer`, `scene.deformation`, `scene.blce`, `scene.gaussian_model`, `helper_model` importable by
 (i) injecting stub modules for the absent packages (cv2, plyfile, simple_knn, torchdiffeq, pytorch3d, tkinter,
     mmengine ...), with `gsplat.rendering` provided by the CPU oracle oracle/gsplat_torch.py,
 (ii) registering `scene` as a bare namespace package (its __init__ pulls in the dataset readers),
 (iii) rewriting device="cuda" to "cpu" with a TorchFunctionMode and making Tensor.cuda()/Module.cuda() no-ops.
Nothing of the reference is copied: it is imported from where it lies and only its OUTPUTS are stored as
fixtures by make_golden.py.  /root/reference does not exist on the GPU box, so nothing here runs there.
"""
from __future__ import annotations

import importlib
import os
import sys
import types

import torch
from torch.overrides import TorchFunctionMode

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "gaussian_renderer"))


def _is_cuda(d) -> bool:
    return (isinstance(d, str) and d.startswith("cuda")) or (isinstance(d, torch.device) and d.type == "cuda")


class CudaToCpu(TorchFunctionMode):
    def __torch_function__(self, func, types_, args=(), kwargs=None):
        kwargs = dict(kwargs or {})
        if "device" in kwargs and _is_cuda(kwargs["device"]):
            kwargs["device"] = "cpu"
        if any(_is_cuda(a) for a in args):
            args = tuple("cpu" if _is_cuda(a) else a for a in args)
        return func(*args, **kwargs)


def _euler_odeint(func, y0, t, method="euler", **_):
    """torchdiffeq.odeint(method='euler') on the given grid: fixed-step explicit Euler
    (the only form the reference uses, /root/reference/scene/blce.py:307-308)."""
    assert method == "euler"
    ys = [y0]
    y = y0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        y = y + dt * func(t[i], y)
        ys.append(y)
    return torch.stack(ys, 0)


_installed = False


def install():
    """Idempotent.  After this, `import gaussian_renderer` etc. resolve to /root/reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("/root/reference is not present")
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from oracle import gsplat_torch

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("cv2")
    stub("tkinter", W="w")
    stub("plyfile", PlyData=object, PlyElement=object)
    stub("simple_knn")
    stub("simple_knn._C", distCUDA2=None)
    stub("torchdiffeq", odeint=_euler_odeint, odeint_adjoint=_euler_odeint)
    stub("pytorch3d")
    stub("pytorch3d.transforms", quaternion_to_matrix=None, matrix_to_quaternion=None,
         axis_angle_to_matrix=None, matrix_to_axis_angle=None, se3_exp_map=None, se3_log_map=None)
    stub("mmengine")
    stub("lpips")
    stub("pytorch3d.ops", ball_query=None)
    stub("open3d")
    stub("gsplat")
    stub("gsplat.rendering", rasterization=gsplat_torch.rasterization,
         fully_fused_projection=gsplat_torch.fully_fused_projection)
    sys.modules["gsplat"].rendering = sys.modules["gsplat.rendering"]

    # `scene` as a namespace package (skip scene/__init__.py -> dataset readers -> cv2 ...)
    scene = types.ModuleType("scene")
    scene.__path__ = [os.path.join(REF, "scene")]
    sys.modules["scene"] = scene
    # scene.cameras drags in dycheck_geometry -> ffmpeg/jax...; blce.py only needs the Camera NAME
    stub("scene.cameras", Camera=type("Camera", (), {}))
    # the reference's top-level packages must win over same-named ones in this repo
    sys.path.insert(0, REF)

    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    _installed = True


def ref_import(name: str):
    install()
    with CudaToCpu():
        return importlib.import_module(name)


class Args:
    """ModelHiddenParams with the seesaw config merged (arguments/__init__.py:77-108 defaults,
    arguments/stereo/default.py:1-14, arguments/stereo/seesaw.py:3-10), as train.py builds them."""

    def __init__(self):
        self.net_width = 128
        self.timebase_pe = 4
        self.defor_depth = 1
        self.posebase_pe = 10
        self.scale_rotation_pe = 2
        self.opacity_pe = 2
        self.timenet_width = 64
        self.timenet_output = 32
        self.bounds = 1.6
        self.plane_tv_weight = 0.0002
        self.time_smoothness_weight = 0.001
        self.l1_time_planes = 0.0001
        self.kplanes_config = {"grid_dimensions": 2, "input_coordinate_dim": 4, "output_coordinate_dim": 32,
                               "resolution": [64, 64, 64, 12]}
        self.multires = [1, 2, 4]
        self.no_dx = False
        self.no_grid = False
        self.no_ds = False
        self.no_dr = False
        self.no_do = True
        self.no_dshs = True
        self.empty_voxel = False
        self.grid_pe = 0
        self.static_mlp = False
        self.apply_rotation = False
        self.render_process = True
