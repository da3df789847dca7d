"""mobgs_amd -- MI355X-native (gfx950) Gaussian-splatting render path for MoBGS.

Layers:
  B3  csrc/libmobgs_hip.so   hand-written HIP kernels behind the C ABI of include/mobgs_hip.h
  B2  mobgs_amd.rendering    drop-in for gsplat.rendering (rasterization, fully_fused_projection)
  B1  mobgs_amd.gaussian_renderer   drop-in for the reference's gaussian_renderer (render, get_flow, ...)
"""
__version__ = "0.1.0"
