#!/usr/bin/env python3
"""Per-iteration launch shape of pm_step: the full config-3 schedule (16 views of
1920x1080, 7x7, S=4, 8 x (2 + 8)) with ONE strip height and ONE residency cap for all launches, for
every (rows, workgroups per CU) of a grid; per-launch device times (amvs_get_step_times) averaged by
iteration and by kind (propagation / refinement).  Also checks that the maps do not depend on the
shape.  Prints one table line per combination and the per-iteration optimum.

    python tools/step_shape.py [--rows 8,12,16,24,32,48,64] [--caps 3,4,5,6] [--views 16] [--mode fast]
"""
import argparse
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="8,12,16,24,32,48,64")
    ap.add_argument("--caps", default="3,4,5,6")
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--scene-views", type=int, default=0)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--mode", default="fast")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--json", default="")
    ap.add_argument("--schedule", default="auto")
    ap.add_argument("--views-per-launch", type=int, default=0)
    args = ap.parse_args()
    import torch

    import amvs
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    H, W = args.height, args.width
    n_scene = args.scene_views or args.views
    sc = make_scene(n_scene, H, W, seed=1234, device="cuda")
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    eng = amvs.Engine(H, W, n_scene, sc.camera.K.astype(np.float32), mode=args.mode)
    for i in ids:
        g = (np.round(sc.grays[i] * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
        eng.set_view(i, g, sc.poses[i].R, sc.poses[i].t)
    dev = torch.device("cuda", 0)
    n = args.views
    depth = torch.empty((n, H * W), dtype=torch.float32, device=dev)
    normal = torch.empty((n, 3 * H * W), dtype=torch.float32, device=dev)
    conf = torch.empty((n, H * W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    p = make_pm_params(7, 8, 8, sc.depth_min, sc.depth_max, schedule=args.schedule, views_per_launch=args.views_per_launch)
    eng.set_step_timing(True)
    iters, per_it = 8, 10
    results = {}
    digest0 = None
    for rows in [int(x) for x in args.rows.split(",")]:
        for cap in [int(x) for x in args.caps.split(",")]:
            eng.set_step_tuning([(rows, rows)], [(cap, cap)])
            acc = np.zeros(iters * per_it)
            for rep in range(args.reps + 1):
                eng.patchmatch_device(ids[:n], sources[:n], p, 42, depth.data_ptr(), normal.data_ptr(), conf.data_ptr())
                eng.sync()
                t = eng.step_times()
                # groups of views run one after the other, each through the whole schedule: the time of a
                # step is the sum over the groups (= per launch of all `n` views, as in the one-launch case)
                assert len(t) % (iters * per_it) == 0, len(t)
                t = np.asarray(t).reshape(-1, iters * per_it).sum(axis=0)
                if rep > 0:
                    acc += t
            acc /= args.reps
            dg = hashlib.sha1(depth.cpu().numpy().tobytes() + normal.cpu().numpy().tobytes() + conf.cpu().numpy().tobytes()).hexdigest()
            digest0 = digest0 or dg
            assert dg == digest0, f"maps depend on the launch shape (rows {rows}, cap {cap})"
            by = acc.reshape(iters, per_it)
            results[(rows, cap)] = by
            print(f"rows {rows:3d} cap {cap}: mean {acc.mean():.4f} ms | prop by iter " +
                  " ".join(f"{by[i, :2].mean():.3f}" for i in range(iters)) + " | refine by iter " +
                  " ".join(f"{by[i, 2:].mean():.3f}" for i in range(iters)), flush=True)
    print("\nper-iteration optimum (rows, cap, ms):")
    best_total = 0.0
    table = {"prop": [], "refine": []}
    for i in range(iters):
        for kind, sl in (("prop", slice(0, 2)), ("refine", slice(2, per_it))):
            k, v = min(((k, v[i, sl].mean()) for k, v in results.items()), key=lambda kv: kv[1])
            best_total += v * (2 if kind == "prop" else per_it - 2)
            table[kind].append((k[0], k[1], round(float(v), 4)))
            print(f"  iter {i} {kind:6s}: rows {k[0]:3d} cap {k[1]}  {v:.4f} ms")
    uni = min(results.items(), key=lambda kv: kv[1].mean())
    print(f"best uniform shape: rows {uni[0][0]} cap {uni[0][1]}: {uni[1].mean():.4f} ms per launch; "
          f"per-iteration optimum: {best_total / (iters * per_it):.4f} ms per launch")
    if args.json:
        with open(args.json, "w") as f:
            json.dump({"grid": {f"{k[0]}x{k[1]}": v.tolist() for k, v in results.items()}, "best": table}, f)
    eng.close()


if __name__ == "__main__":
    main()
