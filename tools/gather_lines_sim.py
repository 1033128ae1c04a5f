#!/usr/bin/env python3
"""How many distinct 128-byte lines does one 64-lane gather of pm_step touch, and could a memory layout
lower that?  (CPU only, NumPy; the geometry of bench.py's scene: 16 views of 1920x1080 on a 10-degree arc.)

A wave's 64 lanes are 64 adjacent pixels of a reference row, each with its own depth hypothesis (log-uniform
in [depth_min, depth_max], as after pm_init); the gather fetches one dword of the 2-byte row-pair map per
lane.  Counted per (source, reference row, strip): distinct lines under
  plain  -- the library's layout (row-major, 64 texels per line);
  shear  -- columns in groups of g texels, each group shifted vertically by round(m g j) rows so that a line
            follows the epipolar slope m of the strip (an idealised per-strip slope: an upper bound on what
            any sheared copy per (reference, source) pair could achieve);
  cells  -- the samples of a row fill a 2-D parallelogram (the image of the row at one depth has about twice
            the slope of the epipolar lines): its texel area (+ one row) in units of 64 texels -- the number of
            lines the region spans whatever their shape (the 64 samples hit fewer when it exceeds ~40).
Result (printed): plain 17.9 lines per gather on average over the image (8 on the central rows of the +-10
degree sources, 29-31 at the top and bottom rows of the +-20 degree ones), sheared layouts 17.9-20.2 -- no
layout of the 2-byte texels flattens the bands; tools/l1_share.hip prices such gathers on the hardware.
"""
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from amvs.synthetic import arc_poses
    H, W, n = 1080, 1920, 16
    f = 0.8 * W
    K = np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]])
    Ki = np.linalg.inv(K)
    poses = arc_poses(n)
    dmin, dmax = 0.6 * 5, 1.6 * 5
    rng = np.random.default_rng(0)

    def compose(ref, src):
        Rr, tr = poses[ref].R, poses[ref].t
        Rs, ts = poses[src].R, poses[src].t
        Rrel = Rs @ Rr.T
        return K @ Rrel @ Ki, K @ (ts - Rrel @ tr)

    def project(M, b, x, y, d):
        p = d * (M @ np.stack([x, y, np.ones_like(x)])) + b[:, None]
        return p[0] / p[2], p[1] / p[2]

    def lines(u, v, m=None, g=7):
        xi = np.clip(np.floor(u), -2, W).astype(int) + 2
        yi = np.clip(np.floor(v), -2, H).astype(int) + 2
        if m is None:
            addr = yi * 2 * (W + 4) + 2 * xi
        else:
            j = xi // g
            S = np.round(m * g * j).astype(int)
            addr = ((yi - S + 400) * ((W + 4 + g - 1) // g) + j) * (g + 1) * 2 + (xi - g * j) * 2
        return len(np.unique(addr // 128))

    ref = 8
    print("source  rows            plain   shear g=3   g=7   g=15   cells")
    tot = np.zeros(5)
    cnt = 0
    for src in (7, 9, 6, 10):
        M, b = compose(ref, src)
        for band, ys in (("centre (440..640)", range(440, 641, 50)), ("middle", list(range(200, 401, 50)) + list(range(680, 881, 50))),
                         ("top / bottom", list(range(0, 151, 50)) + list(range(930, 1080, 49)))):
            acc = np.zeros(5)
            k = 0
            for y in ys:
                for x0 in range(0, W - 64, 58 * 4):
                    x = np.arange(x0, x0 + 64).astype(float)
                    yy = np.full(64, float(y))
                    d = np.exp(rng.uniform(np.log(dmin), np.log(dmax), 64))
                    u, v = project(M, b, x, yy, d)
                    xm = np.array([x0 + 32.0])
                    u0, v0 = project(M, b, xm, yy[:1], np.array([dmin]))
                    u1, v1 = project(M, b, xm, yy[:1], np.array([dmax]))
                    m = float(((v1 - v0) / (u1 - u0))[0])
                    # area of the parallelogram the samples can fall into: its four corners
                    cu, cv = project(M, b, np.array([x[0], x[-1], x[-1], x[0]]), np.full(4, float(y)), np.array([dmin, dmin, dmax, dmax]))
                    area = 0.5 * abs(sum(cu[i] * cv[(i + 1) % 4] - cu[(i + 1) % 4] * cv[i] for i in range(4)))
                    floor = min(64.0, max(1.0, (area + abs(cu[2] - cu[1])) / 64.0))      # (+ one row: the band is at least a texel high)
                    acc += [lines(u, v), lines(u, v, m, 3), lines(u, v, m, 7), lines(u, v, m, 15), floor]
                    k += 1
            print(f"{src:4d}    {band:18s} {acc[0] / k:5.1f}   {acc[1] / k:5.1f}     {acc[2] / k:5.1f}  {acc[3] / k:5.1f}   {acc[4] / k:5.1f}")
            tot += acc
            cnt += k
    t = tot / cnt
    print(f"all     (equal weight)     {t[0]:5.1f}   {t[1]:5.1f}     {t[2]:5.1f}  {t[3]:5.1f}   {t[4]:5.1f}")


if __name__ == "__main__":
    main()
