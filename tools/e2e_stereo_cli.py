#!/usr/bin/env python3
"""Wall-clock split of DenseStereoReconstructor(scale=0.25).reconstruct at the CLI-default operating point of
bench.py's cli_defaults.stereo record (16 views of 4032x3024 -> 1008x756, 64 planes): where the time goes once the
raw cloud exceeds the 500 k points above which the reference sub-samples at random (dense_stereo.py:449-451)."""
import contextlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amvs  # noqa: E402
from amvs.core import dense_stereo as ds  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402

n_views, H, W, scale = 16, 756, 1008, 0.25
small = make_scene(n_views, H, W, seed=4321, device="cuda")
ids = sorted(small.poses)
images = [{"image": np.ascontiguousarray(np.repeat(np.repeat(small.colors[i], 4, axis=0), 4, axis=1))} for i in ids]
K = small.camera.K.copy()
K[:2] *= 1.0 / scale
cam = amvs.Camera(K=K, dist=np.zeros(5))
marks = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)

    def wrap(*a, **k):
        t = time.perf_counter()
        r = fn(*a, **k)
        marks[label or name] = marks.get(label or name, 0.0) + time.perf_counter() - t
        return r
    setattr(obj, name, wrap)


for name in ("_prepare_images_device", "_sweep_and_backproject", "_filter_and_downsample_device", "_filter_outliers",
             "_voxel_down_sample"):
    fn = getattr(ds.DenseStereoReconstructor, name)

    def make(fn=fn, name=name):
        def wrap(self, *a, **k):
            t = time.perf_counter()
            r = fn(self, *a, **k)
            marks[name] = marks.get(name, 0.0) + time.perf_counter() - t
            return r
        return wrap
    setattr(ds.DenseStereoReconstructor, name, make())
timed(np.random, "choice", "np.random.choice")
from amvs import engine as _eng  # noqa: E402
for name in ("cloud_take", "cloud_knn_mean_distance", "cloud_voxel_downsample", "fetch_cloud", "stereo_backproject_views",
             "stereo_backproject", "plane_sweep_batch"):
    fn = getattr(_eng.Engine, name)

    def make2(fn=fn, name=name):
        def wrap(self, *a, **k):
            t = time.perf_counter()
            r = fn(self, *a, **k)
            marks["eng." + name] = marks.get("eng." + name, 0.0) + time.perf_counter() - t
            return r
        return wrap
    setattr(_eng.Engine, name, make2())
sink = open(os.devnull, "w")
for rep in range(3):
    marks.clear()
    with contextlib.redirect_stdout(sink):
        m = ds.DenseStereoReconstructor(cam, scale=scale)
        t0 = time.perf_counter()
        pts, cols = m.reconstruct(images, small.poses, max_pairs=30)
        dt = time.perf_counter() - t0
    print(f"rep {rep}: total {dt:.3f} s, {len(pts)} points; " + ", ".join(f"{k} {v*1e3:.1f} ms" for k, v in marks.items()), flush=True)
