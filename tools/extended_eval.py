"""Extended PatchMatch mode against ground truth: accuracy per iteration count and time per iteration.
    python tools/extended_eval.py [--views 7 --height 480 --width 640 --iters 1 2 3 4 6]
Prints one JSON line per iteration count: fraction of the middle view's interior pixels within 1 % /
0.2 % of the true depth, the same for the parity mode, and seconds per view."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import amvs  # noqa: E402
from amvs.core.mvs_patchmatch import PatchMatchMVS  # noqa: E402
from amvs.synthetic import make_scene  # noqa: E402


def run(sc, iters, extended, quiet=True):
    pm = PatchMatchMVS(sc.camera, scale=1.0, patch_size=7, num_iterations=iters, num_samples=4, min_views=3,
                       seed=11, device=0, extended=extended)
    pm._estimate_depth_range = lambda poses, sparse: None
    pm.depth_min, pm.depth_max = sc.depth_min, sc.depth_max
    keep = {}
    orig = pm._fuse_filter_resident

    def spy(maps, images, poses):
        keep["maps"] = maps
        return orig(maps, images, poses)
    pm._fuse_filter_resident = spy
    out = open(os.devnull, "w") if quiet else sys.stdout
    stdout, sys.stdout = sys.stdout, out
    try:
        t0 = time.time()
        pts, _ = pm.reconstruct(sc.images(), dict(sc.poses))
        dt = time.time() - t0
    finally:
        sys.stdout = stdout
    return keep["maps"], pts, dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--views", type=int, default=7)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--iters", type=int, nargs="+", default=[1, 2, 3, 4, 6])
    a = ap.parse_args()
    sc = make_scene(a.views, a.height, a.width, seed=17)
    mid = a.views // 2
    gt = sc.depths[mid][6:-6, 6:-6]
    run(sc, 1, True)                                            # warm-up (library load, allocations)
    for it in a.iters:
        rec = {"iterations": it, "views": a.views, "size": [a.height, a.width]}
        for name, ext in (("extended", True), ("parity", False)):
            maps, pts, dt = run(sc, it, ext)
            H, W = maps.shape
            n = maps.ref_ids.index(mid)
            d = maps.depth.cpu().numpy().reshape(-1, H, W)[n][6:-6, 6:-6]
            err = np.abs(d - gt) / gt
            rec[name] = {"within_1pct": round(float((err <= 0.01).mean()), 4),
                         "within_0.2pct": round(float((err <= 0.002).mean()), 4),
                         "points": int(len(pts)), "seconds_total": round(dt, 3)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
