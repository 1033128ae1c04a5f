#!/usr/bin/env python3
"""Timing split of the stereo path's device post-steps on a synthetic 8-view 720p scene (GPU box)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import amvs  # noqa: E402,F401
from amvs.core import dense_stereo as ds  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402

sc = make_scene(8, 720, 1280, device="cuda")
images = [{"image": np.ascontiguousarray(c[:, :, ::-1])} for c in sc.colors]
rec = ds.DenseStereoReconstructor(sc.camera, scale=1.0)
rec.reconstruct(images, dict(sc.poses))
eng = rec._engine
proc = rec._engine_images
ids = sorted(proc)
K_inv = np.linalg.inv(rec.K_scaled)
for rep in range(3):
    t = [time.time()]
    counts, total = eng.stereo_backproject(np.stack([proc[i]["color"] for i in ids]), K_inv,
                                           [(sc.poses[i].R, sc.poses[i].t) for i in ids], 2.5)
    t.append(time.time())
    mean_d = eng.cloud_knn_mean_distance(total, 20)
    t.append(time.time())
    keep = mean_d < np.mean(mean_d) + 2.0 * np.std(mean_d)
    t.append(time.time())
    m = eng.cloud_voxel_downsample(0.02, keep)
    t.append(time.time())
    pts, cols = eng.fetch_cloud(m)
    t.append(time.time())
    host = eng.knn_mean_distance(eng.fetch_cloud(m)[0], 20)
    t.append(time.time())
    print(f"rep {rep}: {total} pts: backproject {t[1]-t[0]:.4f}, knn(resident) {t[2]-t[1]:.4f}, threshold {t[3]-t[2]:.4f}, "
          f"voxel {t[4]-t[3]:.4f} -> {m}, fetch {t[5]-t[4]:.4f}; knn(host cloud of {m}) {t[6]-t[5]:.4f}")
