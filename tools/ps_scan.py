#!/usr/bin/env python3
"""Plane sweep (BASELINE config 2, fast arithmetic): wall time of one launch by strip rows and planes per
wave (amvs_set_sweep_tuning).

    python tools/ps_scan.py [--rows 32,24,16] [--chunks 0,8,13,16,32]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="60,48,40,32,24")
    ap.add_argument("--chunks", default="0,4,6,8,11,13,16")
    ap.add_argument("--mode", default="fast", choices=["fast", "exact"])
    args = ap.parse_args()
    import torch

    import amvs
    from amvs.synthetic import make_scene
    H, W, n, D, k, S = 720, 1280, 8, 64, 5, 6
    sc = make_scene(n, H, W, seed=1234, device="cuda")
    ids = sorted(sc.poses)
    ds = amvs.DenseStereoReconstructor.__new__(amvs.DenseStereoReconstructor)
    nbrs = [ds._find_neighbors(r, ids, sc.poses, k=S) for r in ids]
    depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, D)).astype(np.float32)
    dev = torch.device("cuda", 0)
    dmap = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    conf = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=args.mode) as eng:
        for i in ids:
            g = (np.round(sc.grays[i] * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
            eng.set_view(i, g, sc.poses[i].R, sc.poses[i].t)
        for rows in [int(x) for x in args.rows.split(",")]:
            for chunk in [int(x) for x in args.chunks.split(",")]:
                eng.set_sweep_tuning(rows, chunk)
                ts = []
                for _ in range(4):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    eng.plane_sweep_device(ids, nbrs, depths, k, 0.8, dmap.data_ptr(), conf.data_ptr())
                    eng.sync()
                    ts.append(time.perf_counter() - t0)
                ms = min(ts[1:]) * 1e3
                print(f"rows {rows} chunk {chunk}: {ms:.3f} ms  {n * H * W * D / ms / 1e6:.1f} G px-hyp/s", flush=True)


if __name__ == "__main__":
    main()
