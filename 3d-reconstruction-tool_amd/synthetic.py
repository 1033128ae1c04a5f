"""Procedural calibrated multi-view scenes.

The reference ships no usable image sets, so tests and bench.py render their own:
a textured height field  Z = amp * sin(fx*X) * cos(fy*Y)  seen by `n_views` cameras
on a circular arc (radius `radius`, `arc_step_deg` apart) that all look at the
origin.  Every view is rendered analytically (ray / surface intersection by
fixed-point iteration), so ground-truth depth maps come with the images.
Rendering uses torch ops on the requested device; it is data generation, not part
of the product path.
"""
import math
from dataclasses import dataclass
from typing import Dict, List

import numpy as np

from .core.camera import Camera, CameraPose


@dataclass
class Scene:
    camera: Camera                  # K at the rendered resolution, zero distortion
    poses: Dict[int, CameraPose]
    grays: List[np.ndarray]         # float32 (H,W) in [0,1]
    colors: List[np.ndarray]        # uint8 (H,W,3) BGR
    depths: List[np.ndarray]        # ground-truth depth along the camera z axis
    depth_min: float
    depth_max: float

    def images(self):
        """The `images` list of dicts SfMPipeline hands to the dense stage (sfm_pipeline.py:116-120)."""
        return [{"image": c} for c in self.colors]


def arc_poses(n_views, radius=5.0, arc_step_deg=10.0):
    poses = {}
    mid = 0.5 * (n_views - 1)
    for i in range(n_views):
        phi = math.radians((i - mid) * arc_step_deg)
        C = np.array([radius * math.sin(phi), 0.0, -radius * math.cos(phi)])
        fwd = -C / np.linalg.norm(C)
        down = np.array([0.0, 1.0, 0.0])
        right = np.cross(down, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        R = np.stack([right, down, fwd])
        poses[i] = CameraPose(R=R, t=-R @ C)
    return poses


def _texture(X, Y, tables, torch):
    """Sum of sinusoids + two octaves of smooth value noise, in [0,1]."""
    val = (0.5 + 0.18 * torch.sin(3.1 * X + 0.7) * torch.cos(2.3 * Y - 0.4)
           + 0.12 * torch.sin(7.9 * X - 1.3 * Y) + 0.08 * torch.cos(13.7 * Y + 4.1 * X))
    for tab, cell, amp in tables:
        n = tab.shape[0]
        gx = (X / cell) % n
        gy = (Y / cell) % n
        x0 = torch.floor(gx)
        y0 = torch.floor(gy)
        fx = gx - x0
        fy = gy - y0
        x0 = x0.long() % n
        y0 = y0.long() % n
        x1 = (x0 + 1) % n
        y1 = (y0 + 1) % n
        v = (tab[y0, x0] * (1 - fx) * (1 - fy) + tab[y0, x1] * fx * (1 - fy)
             + tab[y1, x0] * (1 - fx) * fy + tab[y1, x1] * fx * fy)
        val = val + amp * (v - 0.5)
    return val.clamp(0.0, 1.0)


def make_scene(n_views, H, W, seed=1234, amp=0.15, radius=5.0, arc_step_deg=10.0,
               focal_scale=0.8, device="cpu", depth_lo=0.6, depth_hi=1.6) -> Scene:
    import torch

    dev = torch.device(device)
    rng = np.random.default_rng(seed)
    tables = [(torch.from_numpy(rng.random((256, 256)).astype(np.float32)).to(dev), 0.05, 0.35),
              (torch.from_numpy(rng.random((256, 256)).astype(np.float32)).to(dev), 0.0125, 0.25)]
    f = focal_scale * W
    K = np.array([[f, 0.0, W / 2.0], [0.0, f, H / 2.0], [0.0, 0.0, 1.0]])
    poses = arc_poses(n_views, radius, arc_step_deg)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32),
                            torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    rx = (xs - K[0, 2]) / K[0, 0]
    ry = (ys - K[1, 2]) / K[1, 1]
    sfx, sfy = 1.3, 1.7
    grays, colors, depths = [], [], []
    for i in range(n_views):
        R = torch.from_numpy(poses[i].R.astype(np.float32)).to(dev)
        C = torch.from_numpy(poses[i].center.astype(np.float32)).to(dev)
        # world ray directions d = R^T [rx, ry, 1]
        dx = R[0, 0] * rx + R[1, 0] * ry + R[2, 0]
        dy = R[0, 1] * rx + R[1, 1] * ry + R[2, 1]
        dz = R[0, 2] * rx + R[1, 2] * ry + R[2, 2]
        lam = -C[2] / dz
        for _ in range(12):
            X = C[0] + lam * dx
            Y = C[1] + lam * dy
            lam = (amp * torch.sin(sfx * X) * torch.cos(sfy * Y) - C[2]) / dz
        X = C[0] + lam * dx
        Y = C[1] + lam * dy
        gray = _texture(X, Y, tables, torch)
        grays.append(gray.cpu().numpy().astype(np.float32))
        depths.append(lam.cpu().numpy().astype(np.float32))       # camera-z depth (ray has z = 1)
        g8 = torch.round(gray * 255.0)
        col = torch.stack([(g8 * 0.9).clamp(0, 255), g8, (g8 * 0.8 + 20).clamp(0, 255)], dim=-1)
        colors.append(col.to(torch.uint8).cpu().numpy())
    return Scene(camera=Camera(K=K, dist=np.zeros(5)), poses=poses, grays=grays, colors=colors,
                 depths=depths, depth_min=depth_lo * radius, depth_max=depth_hi * radius)
