#!/usr/bin/env python3
"""Wall-clock split of PatchMatchMVS.reconstruct on a synthetic scene (run on the GPU box):
image preparation, sweep (upload + kernels + download), fusion + filter."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import amvs  # noqa: E402
from amvs.core import mvs_patchmatch as mp  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402

n_views = int(sys.argv[1]) if len(sys.argv) > 1 else 16
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
scale = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
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


for name in ("_prepare_images", "_sweep", "_sweep_resident", "_fuse_filter_device", "_fuse_filter_resident",
             "_estimate_depth_range", "_prepare_images_device"):
    timed(mp.PatchMatchMVS, name)
for rep in range(4):
    marks.clear()
    m = mp.PatchMatchMVS(sc.camera, scale=scale, patch_size=7, num_iterations=8, num_samples=8)
    t0 = time.time()
    pts, cols = m.reconstruct(images, poses)
    total = time.time() - t0
    print(f"E2E rep {rep}: total {total:.3f} s, points {len(pts)}, " +
          ", ".join(f"{k} {v:.3f}" for k, v in marks.items()))
