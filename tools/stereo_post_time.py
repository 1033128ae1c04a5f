#!/usr/bin/env python3
"""Timing split of the stereo path's device post-steps (GPU box): back-projection of the resident sweep maps,
the reference's random sub-sample above 500 000 points on the device (amvs_cloud_take), neighbour statistic,
threshold, voxel grid, fetch.

    python tools/stereo_post_time.py [n_views H W]          default: 16 756 1008 (the CLI-default shape: a 2 M-point cloud)
"""
import contextlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amvs  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402

n, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 756, 1008)
sc = make_scene(n, H, W, seed=4321, device="cuda")
ids = sorted(sc.poses)
ds = amvs.DenseStereoReconstructor(sc.camera, scale=1.0)
with contextlib.redirect_stdout(open(os.devnull, "w")):
    ds.reconstruct(sc.images(), sc.poses)
eng = ds._engine
K_inv = np.linalg.inv(ds.K_scaled)
poses = [(sc.poses[i].R, sc.poses[i].t) for i in ids]
slots = [ds._slot[i] for i in ids]
for rep in range(4):
    t = [time.perf_counter()]
    counts, total = eng.stereo_backproject_views(slots, K_inv, poses, ds.min_views - 0.5)
    t.append(time.perf_counter())
    m0 = total
    if total > 500000:
        chosen = ds._draw_without_replacement(total, 500000)      # np.random.choice's draw, in kept buffers (the class)
        t.append(time.perf_counter())
        m0 = eng.cloud_take(chosen)
    else:
        t.append(time.perf_counter())
    t.append(time.perf_counter())
    md = eng.cloud_knn_mean_distance(m0, 20)
    t.append(time.perf_counter())
    keep = md < md.mean() + 2 * md.std()
    t.append(time.perf_counter())
    m = eng.cloud_voxel_downsample(0.02, keep)
    t.append(time.perf_counter())
    p, c = eng.fetch_cloud(m)
    t.append(time.perf_counter())
    d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
    print(f"rep {rep}: {total} raw points: backproject {d[0]:.1f} ms, draw {d[1]:.1f}, take {d[2]:.1f}, knn({m0}) {d[3]:.1f}, "
          f"threshold {d[4]:.1f}, voxel {d[5]:.1f} -> {m}, fetch {d[6]:.1f}", flush=True)
