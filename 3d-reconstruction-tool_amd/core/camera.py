"""Pinhole camera and pose types used on the dense-reconstruction boundary.

Same public surface as the reference's src/core/camera.py (Camera :11-75,
CameraPose :78-108, load_calibration :111-139) so callers can pass either.
Convention: X_cam = R @ X_world + t.
"""
from dataclasses import dataclass
from pathlib import Path

import numpy as np


@dataclass
class Camera:
    """Intrinsics K (3x3, [[fx,0,cx],[0,fy,cy],[0,0,1]]) and distortion (k1,k2,p1,p2,k3)."""
    K: np.ndarray
    dist: np.ndarray

    fx = property(lambda self: self.K[0, 0])
    fy = property(lambda self: self.K[1, 1])
    cx = property(lambda self: self.K[0, 2])
    cy = property(lambda self: self.K[1, 2])

    def project(self, points_3d: np.ndarray) -> np.ndarray:
        """Camera-frame points (N,3) -> pixel coordinates (N,2)."""
        xy = points_3d[:, :2] / points_3d[:, 2:3]
        return np.column_stack([self.fx * xy[:, 0] + self.cx, self.fy * xy[:, 1] + self.cy])

    def unproject(self, points_2d: np.ndarray, depth: float = 1.0) -> np.ndarray:
        """Pixel coordinates (N,2) -> camera-frame points at the given depth (N,3)."""
        xn = (points_2d[:, 0] - self.cx) / self.fx
        yn = (points_2d[:, 1] - self.cy) / self.fy
        return np.column_stack([xn * depth, yn * depth, np.full(len(points_2d), float(depth))])


@dataclass
class CameraPose:
    """World-to-camera rigid transform."""
    R: np.ndarray
    t: np.ndarray

    @property
    def center(self) -> np.ndarray:
        return -self.R.T @ self.t.ravel()

    @property
    def projection_matrix(self) -> np.ndarray:
        return np.hstack([self.R, self.t.reshape(3, 1)])

    def transform_points(self, points_world: np.ndarray) -> np.ndarray:
        return (self.R @ points_world.T).T + self.t.ravel()

    @staticmethod
    def identity() -> "CameraPose":
        return CameraPose(R=np.eye(3), t=np.zeros(3))


def load_calibration(calibration_path: str) -> Camera:
    """Read `mtx` / `dist` from a calibration .npz (reference camera.py:111-139)."""
    path = Path(calibration_path)
    if not path.exists():
        raise FileNotFoundError(f"Calibration file not found: {path}")
    with np.load(str(path)) as data:
        K = data["mtx"].astype(np.float64)
        dist = data["dist"].astype(np.float64).ravel()
    if dist.size < 5:
        dist = np.pad(dist, (0, 5 - dist.size))
    print(f"Loaded calibration from {path.name}")
    print(f"  Focal length: fx={K[0,0]:.1f}, fy={K[1,1]:.1f}")
    print(f"  Principal point: cx={K[0,2]:.1f}, cy={K[1,2]:.1f}")
    return Camera(K=K, dist=dist)
