"""Read-side camera contract of the render path.

render()/get_flow() only read these attributes of the reference's `scene.cameras.Camera`
(/root/reference/scene/cameras.py:18-151): world_view_transform (W2C, stored TRANSPOSED), K, time, max_time,
image_width, image_height, cam_ray [1,6,H,W] and get_pixels().  `PinholeCamera` provides exactly those for
synthetic scenes, tests and the benchmark; any object with the same attributes (e.g. the reference's own Camera)
can be passed to mobgs_amd.gaussian_renderer.render instead.
"""
from __future__ import annotations

import numpy as np
import torch


class PinholeCamera:
    def __init__(self, width: int, height: int, K: torch.Tensor, w2c: torch.Tensor, time: float = 0.0,
                 max_time: int = 1, device="cpu"):
        self.image_width = int(width)
        self.image_height = int(height)
        self.K = K.to(device=device, dtype=torch.float32)
        self.time = float(time)
        self.max_time = max_time
        w2c = w2c.to(device=device, dtype=torch.float32)
        self.world_view_transform = w2c.transpose(0, 1)  # the reference stores the transpose (:121-130)
        self._w2c = w2c
        self._ray = None
        # pinhole parameters for in-kernel ray generation (mobgs_amd.ops.decode), computed once:
        # [fx, fy, cx, cy] and c2w [3,4]
        self.ray_intrinsics = torch.stack([self.K[0, 0], self.K[1, 1], self.K[0, 2], self.K[1, 2]])
        self.ray_c2w = torch.inverse(w2c)[:3, :].contiguous()

    @property
    def cam_ray(self):
        """[1,6,H,W] ray map, built on first use (33 MB at 1352x1014; render() does not need it)."""
        if self._ray is None:
            self._ray = self.build_cam_ray(self.image_width, self.image_height, self.K, self._w2c)
        return self._ray

    @staticmethod
    def build_cam_ray(width, height, K, w2c):
        """[1,6,H,W]: camera centre (3) + unit view direction through each pixel CENTRE (3), world frame
        (/root/reference/scene/cameras.py:132-146, :206-213; dycheck pixels_to_viewdirs)."""
        return PinholeCamera.build_cam_ray_c2w(width, height, K, torch.inverse(w2c)[:3, :])

    @staticmethod
    def build_cam_ray_c2w(width, height, K, c2w):
        """Same map from the camera-to-world matrix c2w [3,4] (or [4,4]): direction = normalise(R_c2w @ local),
        origin = c2w[:3, 3] -- the function the decoder kernel evaluates in registers (csrc/decoder.hip)."""
        dev = K.device
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=dev),
                                torch.arange(width, dtype=torch.float32, device=dev), indexing="ij")
        x = (xs + 0.5 - K[0, 2]) / K[0, 0]
        y = (ys + 0.5 - K[1, 2]) / K[1, 1]
        local = torch.stack([x, y, torch.ones_like(x)], dim=-1)
        dirs = local @ c2w[:3, :3].transpose(0, 1)
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        ray = torch.cat([c2w[:3, 3].expand_as(dirs), dirs], dim=-1)
        return ray.permute(2, 0, 1).unsqueeze(0).contiguous()

    def get_pixels(self, image_size_x, image_size_y, use_center=None):
        xx, yy = np.meshgrid(np.arange(image_size_x, dtype=np.float32), np.arange(image_size_y, dtype=np.float32))
        return np.stack([xx, yy], axis=-1) + (0.5 if use_center else 0)

    def to(self, device):
        c = object.__new__(PinholeCamera)
        c.__dict__.update(self.__dict__)
        for k in ("K", "world_view_transform", "_w2c", "ray_intrinsics", "ray_c2w"):
            setattr(c, k, getattr(self, k).to(device))
        c._ray = None if self._ray is None else self._ray.to(device)
        return c
