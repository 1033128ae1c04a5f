#!/usr/bin/env python3
"""Randomised check of amvs_knn_mean_distance against scikit-learn (the reference's own expression,
dense_stereo.py:456-460): random clouds -- sheets, blobs of very different density, lines, uniform volumes, exact
duplicates, far outliers, float32-valued coordinates --, sizes from 45 to 250 000, every compiled neighbour count.
Bit equality of the float64 means is required (not part of the test suite; run on the GPU box).

    python tools/fuzz_knn.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import amvs  # noqa: E402
from sklearn.neighbors import NearestNeighbors  # noqa: E402


def cloud(rng):
    parts = []
    for _ in range(int(rng.integers(1, 5))):
        kind = rng.choice(["sheet", "blob", "line", "volume", "shell"])
        m = int(10 ** rng.uniform(1.7, 5.0))
        centre = rng.uniform(-5, 5, 3)
        scale = 10 ** rng.uniform(-2.5, 1.2)
        if kind == "sheet":
            xy = rng.uniform(-1, 1, (m, 2))
            p = np.column_stack([xy, 0.2 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1]) + rng.normal(0, 10 ** rng.uniform(-4, -1.5), m)])
        elif kind == "blob":
            p = rng.normal(0, 1, (m, 3)) * rng.uniform(0.05, 1.0, 3)
        elif kind == "line":
            p = np.zeros((m, 3))
            p[:, int(rng.integers(0, 3))] = rng.uniform(-1, 1, m)
        elif kind == "volume":
            p = rng.uniform(-1, 1, (m, 3))
        else:
            v = rng.normal(size=(m, 3))
            p = v / np.linalg.norm(v, axis=1, keepdims=True)
        parts.append(p * scale + centre)
    pts = np.vstack(parts)
    if rng.random() < 0.4:                                   # exact duplicates
        src = rng.integers(0, len(pts), max(1, len(pts) // 50))
        pts[rng.integers(0, len(pts), len(src))] = pts[src]
    if rng.random() < 0.5:                                   # far outliers
        pts = np.vstack([pts, rng.uniform(-1, 1, (int(rng.integers(1, 400)), 3)) * 10 ** rng.uniform(1, 3)])
    if rng.random() < 0.3:
        pts = pts.astype(np.float32).astype(np.float64)
    if len(pts) > 250000:
        pts = pts[rng.choice(len(pts), 250000, replace=False)]
    return np.ascontiguousarray(pts[rng.permutation(len(pts))])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t_dev = t_ref = 0.0
    with amvs.Engine(8, 8, 1, np.eye(3, dtype=np.float32)) as eng:
        for case in range(args.cases):
            pts = cloud(rng)
            k = int(rng.choice([8, 10, 16, 20, 32]))
            if len(pts) < 2 * k + 2:
                continue
            t = time.perf_counter()
            got = eng.knn_mean_distance(pts, k)
            t_dev += time.perf_counter() - t
            t = time.perf_counter()
            d, _ = NearestNeighbors(n_neighbors=k).fit(pts).kneighbors(pts)
            t_ref += time.perf_counter() - t
            want = np.mean(d[:, 1:], axis=1)
            bad = got != want
            if bad.any():
                print(f"case {case}: n = {len(pts)}, k = {k}: {int(bad.sum())} means differ (max |diff| {np.abs(got - want).max():.3e})")
                return 1
            if case % 10 == 9:
                print(f"{case + 1} cases ok (device {t_dev:.2f} s, scikit-learn {t_ref:.1f} s so far)", flush=True)
    print(f"knn fuzz ok: {args.cases} cases (device {t_dev:.2f} s, scikit-learn {t_ref:.1f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
