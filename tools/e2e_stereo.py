#!/usr/bin/env python3
"""Wall-clock split of DenseStereoReconstructor.reconstruct on a synthetic scene (GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import amvs  # noqa: E402,F401
from amvs.core import dense_stereo as ds  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 8
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1280, 720)
sc = make_scene(n_views, H, W, device="cuda")
images = [{"image": np.ascontiguousarray(c[:, :, ::-1])} for c in sc.colors]
poses = dict(sc.poses) if isinstance(sc.poses, dict) else {i: p for i, p in enumerate(sc.poses)}
marks = {}


def timed(cls, name):
    fn = getattr(cls, name)

    def wrap(self, *a, **k):
        t = time.time()
        r = fn(self, *a, **k)
        marks[name] = marks.get(name, 0.0) + time.time() - t
        return r
    setattr(cls, name, wrap)


for name in ("_prepare_images", "_prepare_images_device", "_sweep_and_backproject", "_filter_and_downsample_device"):
    timed(ds.DenseStereoReconstructor, name)
for rep in range(4):
    marks.clear()
    m = ds.DenseStereoReconstructor(sc.camera, scale=1.0)
    t0 = time.time()
    pts, cols = m.reconstruct(images, poses, max_pairs=30)
    print(f"E2E rep {rep}: total {time.time() - t0:.3f} s, points {len(pts)}, " +
          ", ".join(f"{k} {v:.3f}" for k, v in marks.items()))
