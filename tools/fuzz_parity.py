#!/usr/bin/env python3
"""Randomised bit-exact parity sweep (run on the GPU box; not part of the test suite): random image
sizes, view counts, source counts, patch sizes, schedules (2-3 iterations, so the negative propagation
offsets of odd iterations run), batch sizes and both arithmetic modes -- every swept view against the
CPU oracle.  Prints one line per case; exits non-zero on the first mismatch.

    python tools/fuzz_parity.py [--cases 60] [--seed 1]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--sweep", action="store_true", help="plane sweep instead of PatchMatch (random planes incl. "
                                                         "depths behind / near the source cameras, thresholds, strip heights)")
    args = ap.parse_args()
    import amvs
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    oracle.set_threads(16)
    rng = np.random.default_rng(args.seed)
    if args.sweep:
        return fuzz_sweep(args, rng, amvs, make_scene, oracle)
    for case in range(args.cases):
        n = int(rng.integers(3, 8))
        H, W = int(rng.integers(9, 150)), int(rng.integers(9, 200))
        k = int(rng.choice([3, 5, 7, 9, 11, 13, 15, 17, 19, 21, 23, 27, 31]))   # (31: the run-time-k kernels)
        S = int(rng.integers(2, min(n - 1, 6) + 1))
        iters, samples = int(rng.integers(2, 4)), int(rng.integers(1, 4))
        mode = str(rng.choice(["fast", "exact"]))
        vpl = int(rng.choice([0, 1, 2]))
        rows = int(rng.choice([0, 0, 3, 8, 17]))
        sc = make_scene(n, H, W, seed=int(rng.integers(1, 1000)), arc_step_deg=float(rng.choice([4.0, 10.0, 25.0])))
        grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0) for g in sc.grays]
        refs = list(range(n))
        srcs = [[int(j) for j in rng.permutation([j for j in refs if j != r])[:S]] for r in refs]
        K = sc.camera.K.astype(np.float32)
        with amvs.Engine(H, W, n, K, mode=mode) as eng:
            for i in refs:
                eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
            p = make_pm_params(k, iters, samples, sc.depth_min, sc.depth_max, tile_rows=rows, views_per_launch=vpl,
                               schedule=str(rng.choice(["auto", "auto", "paired", "view-major"])))
            seed = int(rng.integers(0, 2 ** 31))
            d, nrm, cf = eng.patchmatch(refs, srcs, p, seed)
        bad = 0
        for r in refs:
            ctx = oracle.ViewContext(K, grays[r], sc.poses[r].R, sc.poses[r].t, [grays[i] for i in srcs[r]],
                                     [sc.poses[i].R for i in srcs[r]], [sc.poses[i].t for i in srcs[r]], k, mode=mode)
            od, on, oc = ctx.patchmatch(iters, samples, sc.depth_min, sc.depth_max, seed, r)
            for a, b in ((d[r], od), (nrm[r], on), (cf[r], oc)):
                same = (a == b) | (np.isnan(a) & np.isnan(b))
                bad += int((~same).sum())
            ctx.close()
        print(f"case {case}: {n} views {W}x{H} k={k} S={S} {iters}x(2+{samples}) {mode} vpl={vpl} rows={rows} {p.schedule}: "
              f"{'ok' if bad == 0 else str(bad) + ' ELEMENTS DIFFER'}", flush=True)
        if bad:
            sys.exit(1)
    print("fuzz ok:", args.cases, "cases")
    report_index_checks()


def report_index_checks():
    """With an index-checked library (AMVS_LIB=build/variants/libamvs_check.so) the random shapes also went through
    the extent checks of every data-dependent global index: report, and fail on a violation."""
    from amvs import _lib
    if _lib.index_checks_enabled():
        count, tu, line, index, extent = _lib.index_check()
        print(f"index-checked build: {count} out-of-range accesses" + (f" (first: unit {tu} line {line}, index {index}, extent {extent})" if count else ""))
        if count:
            sys.exit(2)


def fuzz_sweep(args, rng, amvs, make_scene, oracle):
    for case in range(args.cases):
        n = int(rng.integers(3, 8))
        H, W = int(rng.integers(9, 150)), int(rng.integers(9, 200))
        k = int(rng.choice([3, 5, 7, 9, 11, 13, 17, 21, 25, 31]))
        S = int(rng.integers(2, min(n - 1, 6) + 1))
        D = int(rng.integers(1, 40))
        mode = str(rng.choice(["fast", "exact"]))
        thresh = float(rng.choice([0.8, 0.5, 0.0, -0.3, 0.3, 0.97, 0.0005]))
        sc = make_scene(n, H, W, seed=int(rng.integers(1, 1000)), arc_step_deg=float(rng.choice([4.0, 10.0, 40.0])))
        grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0) for g in sc.grays]
        # planes from well in front of the scene to beyond it (wide arcs put some of them behind a source)
        depths = (1.0 / np.linspace(1 / (sc.depth_max * 3), 1 / (sc.depth_min * 0.2), D)).astype(np.float32)
        refs = list(range(n))
        K = sc.camera.K.astype(np.float32)
        bad = 0
        with amvs.Engine(H, W, n, K, mode=mode) as eng:
            for i in refs:
                eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
            eng.set_sweep_tuning(int(rng.choice([0, 0, 5, 13, 32, 47, 64])), int(rng.choice([0, 0, 1, 3, 7, 32, 40])))
            for r in refs[:3]:
                nb = [int(j) for j in rng.permutation([j for j in refs if j != r])[:S]]
                dm, cf = eng.plane_sweep(r, nb, depths, k, thresh)
                ctx = oracle.ViewContext(K, grays[r], sc.poses[r].R, sc.poses[r].t, [grays[i] for i in nb],
                                         [sc.poses[i].R for i in nb], [sc.poses[i].t for i in nb], k, mode=mode)
                od, oc = ctx.plane_sweep(depths, thresh)
                bad += int((dm != od).sum()) + int((cf != oc).sum())
                ctx.close()
        print(f"sweep case {case}: {n} views {W}x{H} k={k} S={S} D={D} t={thresh} {mode}: "
              f"{'ok' if bad == 0 else str(bad) + ' ELEMENTS DIFFER'}", flush=True)
        if bad:
            sys.exit(1)
    print("sweep fuzz ok:", args.cases, "cases")
    report_index_checks()


if __name__ == "__main__":
    main()
