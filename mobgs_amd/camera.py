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
        self.cam_ray = self.build_cam_ray(self.image_width, self.image_height, self.K, w2c)

    @staticmethod
    def build_cam_ray(width, height, K, w2c):
        """[1,6,H,W]: camera centre (3) + unit view direction through each pixel CENTRE (3), world frame
        (/root/reference/scene/cameras.py:132-146, :206-213; dycheck pixels_to_viewdirs)."""
        dev = K.device
        ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float32, device=dev),
                                torch.arange(width, dtype=torch.float32, device=dev), indexing="ij")
        x = (xs + 0.5 - K[0, 2]) / K[0, 0]
        y = (ys + 0.5 - K[1, 2]) / K[1, 1]
        local = torch.stack([x, y, torch.ones_like(x)], dim=-1)
        R = w2c[:3, :3]
        dirs = local @ R  # R^T applied to each row: camera -> world
        dirs = dirs / dirs.norm(dim=-1, keepdim=True)
        centre = torch.inverse(w2c)[:3, 3]
        ray = torch.cat([centre.expand_as(dirs), dirs], dim=-1)
        return ray.permute(2, 0, 1).unsqueeze(0).contiguous()

    def get_pixels(self, image_size_x, image_size_y, use_center=None):
        xx, yy = np.meshgrid(np.arange(image_size_x, dtype=np.float32), np.arange(image_size_y, dtype=np.float32))
        return np.stack([xx, yy], axis=-1) + (0.5 if use_center else 0)

    def to(self, device):
        c = object.__new__(PinholeCamera)
        c.__dict__.update(self.__dict__)
        for k in ("K", "world_view_transform", "cam_ray"):
            setattr(c, k, getattr(self, k).to(device))
        return c
