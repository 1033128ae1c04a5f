"""Boundary types of the dense stage: pinhole intrinsics and world-to-camera poses.

The dense classes only read `.K` from a camera and `.R`, `.t`, `.center`,
`.transform_points()` from a pose, so objects of the reference project
(src/core/camera.py:11-108) can be passed straight in; these lightweight
equivalents exist so the backend also runs stand-alone (tests, bench, synthetic
scenes).  Convention everywhere: x_cam = R @ x_world + t.
"""
from pathlib import Path

import numpy as np

_N_DIST = 5      # k1, k2, p1, p2, k3


class Camera:
    """Intrinsic matrix `K` (3x3) plus lens distortion coefficients `dist`."""

    __slots__ = ("K", "dist")

    def __init__(self, K, dist=None):
        self.K = np.asarray(K, dtype=np.float64)
        self.dist = np.zeros(_N_DIST) if dist is None else np.asarray(dist, dtype=np.float64)

    def __repr__(self):
        return f"Camera(fx={self.fx:.2f}, fy={self.fy:.2f}, cx={self.cx:.2f}, cy={self.cy:.2f})"

    def _entry(self, row, col):
        return self.K[row, col]

    fx = property(lambda self: self._entry(0, 0))
    fy = property(lambda self: self._entry(1, 1))
    cx = property(lambda self: self._entry(0, 2))
    cy = property(lambda self: self._entry(1, 2))

    def project(self, points_3d):
        """(N,3) camera-frame points -> (N,2) pixels: perspective divide, then focal / centre."""
        pts = np.asarray(points_3d)
        uv = pts[:, :2] / pts[:, 2:3]
        uv = uv * np.array([self.fx, self.fy]) + np.array([self.cx, self.cy])
        return uv

    def unproject(self, points_2d, depth=1.0):
        """(N,2) pixels -> (N,3) camera-frame points on the plane z = depth."""
        px = np.asarray(points_2d, dtype=np.float64)
        out = np.empty((len(px), 3))
        out[:, 0] = (px[:, 0] - self.cx) / self.fx * depth
        out[:, 1] = (px[:, 1] - self.cy) / self.fy * depth
        out[:, 2] = depth
        return out


class CameraPose:
    """Rigid world-to-camera transform (rotation `R`, translation `t`)."""

    __slots__ = ("R", "t")

    def __init__(self, R, t):
        self.R = R
        self.t = t

    def __repr__(self):
        c = self.center
        return f"CameraPose(center=({c[0]:.3f}, {c[1]:.3f}, {c[2]:.3f}))"

    @classmethod
    def identity(cls):
        return cls(np.eye(3), np.zeros(3))

    @property
    def center(self):
        """Position of the optical centre in the world frame, -R^T t."""
        return -(self.R.T @ np.ravel(self.t))

    @property
    def projection_matrix(self):
        """The 3x4 matrix [R | t]."""
        return np.concatenate([self.R, np.reshape(self.t, (3, 1))], axis=1)

    def transform_points(self, points_world):
        """(N,3) world points -> (N,3) camera-frame points."""
        return (self.R @ points_world.T).T + np.ravel(self.t)


def load_calibration(calibration_path):
    """Camera from a chessboard-calibration archive with arrays `mtx` and `dist`
    (what the reference's calibration tool stores; its loader is camera.py:111-139)."""
    archive = Path(calibration_path)
    if not archive.is_file():
        raise FileNotFoundError(f"Calibration file not found: {archive}")
    with np.load(str(archive)) as npz:
        K = np.array(npz["mtx"], dtype=np.float64)
        dist = np.array(npz["dist"], dtype=np.float64).reshape(-1)
    if dist.size < _N_DIST:
        dist = np.concatenate([dist, np.zeros(_N_DIST - dist.size)])
    cam = Camera(K, dist)
    print(f"Loaded calibration from {archive.name}")
    print(f"  Focal length: fx={cam.fx:.1f}, fy={cam.fy:.1f}")
    print(f"  Principal point: cx={cam.cx:.1f}, cy={cam.cy:.1f}")
    return cam
