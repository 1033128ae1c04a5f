"""Output helpers around the dense stage (reference: src/core/utils.py).

`save_ply` has the reference's signature and writes the same bytes (utils.py:8-37); the per-point
Python `f.write` loop there dominates wall time for multi-million-point clouds, so the formatting
runs in the native library (`amvs_write_ply`, host-only).
"""
import ctypes as C
from pathlib import Path

import numpy as np

from .. import _lib


def save_ply(points: np.ndarray, colors: np.ndarray, output_path: str):
    """Save an (N,3) cloud with (N,3) RGB colours as ASCII PLY."""
    output_path = Path(output_path)
    output_path.parent.mkdir(parents=True, exist_ok=True)
    n = len(points)
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float64).reshape(n, 3))
    cols = np.ascontiguousarray(np.asarray(colors).reshape(n, 3).astype(int).astype(np.int64))
    lib = _lib.load()
    rc = lib.amvs_write_ply(str(output_path).encode(), pts.ctypes.data_as(C.POINTER(C.c_double)),
                            cols.ctypes.data_as(C.POINTER(C.c_int64)), n)
    if rc != 0:
        raise _lib.AmvsError(f"amvs_write_ply failed ({rc}): {lib.amvs_last_error(None).decode()}")
    print(f"Saved {n:,} points to {output_path}")


def compute_scene_bounds(points: np.ndarray) -> dict:
    """Axis-aligned bounds, centre and extent of a cloud (reference utils.py:72-86)."""
    lo, hi = points.min(axis=0), points.max(axis=0)
    return {"min": lo, "max": hi, "center": (lo + hi) / 2, "size": hi - lo}
