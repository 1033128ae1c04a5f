"""Capture golden vectors from the reference implementation (build container only).

    python tests/golden/make_golden.py

Imports the reference's dense modules from /root/reference (see ref_loader.py), drives
them on small synthetic scenes and writes inputs + the reference's outputs to
tests/golden/g*.npz.  The reference never seeds its RNG; "identical RNG streams" are
obtained by replacing torch.rand / torch.randn, for the duration of a call, with
functions that return the tensors of this repository's counter-hash generator
(oracle.rng_fill), in the reference's fixed call order.

Only data (inputs and expected outputs) is stored; no reference source is copied.
"""
import contextlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_loader  # noqa: E402
from oracle import oracle  # noqa: E402

import amvs  # noqa: E402,F401
from amvs.synthetic import make_scene  # noqa: E402

mvs, stereo, refcam = ref_loader.load()
SEED = 42


def quantised_scene(n_views, H, W, seed):
    """Synthetic scene whose gray maps are exactly u8/255 (what real images give)."""
    sc = make_scene(n_views, H, W, seed=seed)
    g8 = [np.round(g * 255.0).astype(np.uint8) for g in sc.grays]
    sc.grays = [g.astype(np.float32) / np.float32(255.0) for g in g8]
    return sc, g8


def scene_arrays(sc, g8):
    ids = sorted(sc.poses)
    return dict(gray_u8=np.stack(g8), color_u8=np.stack(sc.colors),
                K=sc.camera.K.astype(np.float64),
                R=np.stack([sc.poses[i].R for i in ids]).astype(np.float64),
                t=np.stack([sc.poses[i].t for i in ids]).astype(np.float64),
                depth_min=np.float64(sc.depth_min), depth_max=np.float64(sc.depth_max),
                gt_depth=np.stack(sc.depths).astype(np.float32))


class InjectedRng:
    """Serve torch.rand / torch.randn from counter-hash draws in the reference's call order:
    init (draw 0): rand(H,W), randn(H,W), randn(H,W); then per refinement sample (draw d):
    rand(H,W), randn(H,W,3)."""

    def __init__(self, seed, view, first_draw, with_init):
        self.seed, self.view, self.draw = seed, view, first_draw
        self.phase = 0 if with_init else 3
        self.cache = None

    def _get(self, n):
        if self.cache is None:
            self.cache = oracle.rng_fill(self.seed, self.view, self.draw, n)
        return self.cache

    def rand(self, *shape, **kw):
        H, W = shape
        u, _ = self._get(H * W)
        if self.phase >= 3:
            self.phase = 4
        else:
            assert self.phase == 0
            self.phase = 1
        return torch.from_numpy(u.reshape(H, W).copy())

    def randn(self, *shape, **kw):
        _, nz = self._get(shape[0] * shape[1])
        if len(shape) == 2:           # init: two separate randn(H,W) calls
            assert self.phase in (1, 2)
            out = nz[:, self.phase - 1].reshape(shape).copy()
            self.phase += 1
            if self.phase == 3:
                self.draw += 1
                self.cache = None
            return torch.from_numpy(out)
        assert self.phase == 4 and shape[2] == 3
        out = nz.reshape(shape).copy()
        self.phase = 3
        self.draw += 1
        self.cache = None
        return torch.from_numpy(out)


@contextlib.contextmanager
def injected(rng):
    old = torch.rand, torch.randn
    torch.rand, torch.randn = rng.rand, rng.randn
    try:
        yield
    finally:
        torch.rand, torch.randn = old


def ref_pm(sc, patch, iters, samples, min_views=3):
    pm = mvs.PatchMatchMVS(refcam.Camera(K=sc.camera.K.copy(), dist=np.zeros(5)), scale=1.0,
                           patch_size=patch, num_iterations=iters, num_samples=samples,
                           min_views=min_views)
    pm.depth_min, pm.depth_max = sc.depth_min, sc.depth_max
    return pm


def ref_poses(sc):
    return {i: refcam.CameraPose(R=p.R.copy(), t=p.t.copy()) for i, p in sc.poses.items()}


def torch_view(pm, sc, ref, srcs):
    H, W = sc.grays[0].shape
    K = torch.from_numpy(pm.K_scaled.astype(np.float32))
    args = dict(
        ref_gray=torch.from_numpy(sc.grays[ref]),
        src_grays=[torch.from_numpy(sc.grays[i]) for i in srcs],
        K=K, K_inv=torch.inverse(K),
        R_ref=torch.from_numpy(sc.poses[ref].R.astype(np.float32)),
        t_ref=torch.from_numpy(sc.poses[ref].t.astype(np.float32)),
        src_Rs=[torch.from_numpy(sc.poses[i].R.astype(np.float32)) for i in srcs],
        src_ts=[torch.from_numpy(sc.poses[i].t.astype(np.float32)) for i in srcs])
    y, x = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32),
                          indexing="ij")
    args["x_grid"], args["y_grid"] = x, y
    return args


def mixed_depth(sc, ref, seed):
    """Half ground truth, half log-uniform random, a few extreme values (bounds / inf cases)."""
    H, W = sc.grays[0].shape
    rng = np.random.default_rng(seed)
    d = np.exp(rng.uniform(np.log(sc.depth_min), np.log(sc.depth_max), (H, W))).astype(np.float32)
    d[:, W // 2:] = sc.depths[ref][:, W // 2:]
    d[: H // 8, : W // 8] = np.float32(0.05)       # projects far outside the sources
    d[-H // 8:, -W // 8:] = np.float32(400.0)
    return d


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KB")


def main():
    torch.set_num_threads(4)

    # ---- g01 / g02: NCC kernels on plain images (a9, a16) --------------------------------
    rng = np.random.default_rng(7)
    H, W = 48, 64
    a = rng.random((H, W)).astype(np.float32)
    b = (0.6 * a + 0.4 * rng.random((H, W))).astype(np.float32)
    a[10:24, 20:40] = 0.5            # constant region: var1 == 0 (and tiny negative) cases
    b[30:40, 5:30] = 0.25
    out = {}
    for k in (5, 7, 11):
        pm = mvs.PatchMatchMVS(refcam.Camera(K=np.eye(3), dist=np.zeros(5)), patch_size=k)
        out[f"cost_k{k}"] = pm._ncc_cost(torch.from_numpy(a), torch.from_numpy(b)).numpy()
    save("g01_ncc_cost", img1=a, img2=b, **out)
    ds = stereo.DenseStereoReconstructor(refcam.Camera(K=np.eye(3), dist=np.zeros(5)))
    out = {f"ncc_k{k}": ds._compute_ncc_torch(torch.from_numpy(a), torch.from_numpy(b), k).numpy()
           for k in (5, 7)}
    save("g02_stereo_ncc", img1=a, img2=b, **out)

    # ---- scene A: 5 views 64x96 -----------------------------------------------------------
    scA, g8A = quantised_scene(5, 64, 96, seed=11)
    save("scene_a", **scene_arrays(scA, g8A))
    ref, srcs4, srcs2 = 2, [1, 3, 0, 4], [3, 1]
    depth = mixed_depth(scA, ref, 3)

    # g03 patch cost (a8), S=4 and S=2, k=7 and k=11
    out = {}
    for k in (7, 11):
        pm = ref_pm(scA, k, 1, 1)
        for tag, srcs in (("s4", srcs4), ("s2", srcs2)):
            tv = torch_view(pm, scA, ref, srcs)
            out[f"cost_k{k}_{tag}"] = pm._compute_patch_cost(
                tv["ref_gray"], torch.from_numpy(depth), None, tv["src_grays"], tv["K"], tv["K_inv"],
                tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"], tv["x_grid"], tv["y_grid"]).numpy()
    save("g03_patch_cost", ref=ref, srcs4=srcs4, srcs2=srcs2, depth=depth, **out)

    # g07 confidence (a12)
    pm = ref_pm(scA, 7, 1, 1)
    tv = torch_view(pm, scA, ref, srcs4)
    conf = pm._compute_confidence(torch.from_numpy(depth), None, tv["ref_gray"], tv["src_grays"], tv["K"],
                                  tv["K_inv"], tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"],
                                  tv["x_grid"], tv["y_grid"]).numpy()
    save("g07_confidence", ref=ref, srcs=srcs4, patch=7, depth=depth, confidence=conf)

    # state for the step fixtures: init from draw 0, then one evaluation to get finite costs
    Hh, Ww = scA.grays[0].shape
    u0, nz0 = oracle.rng_fill(SEED, ref, 0, Hh * Ww)
    d0, n0, c0 = oracle.init_state(u0.reshape(Hh, Ww), nz0[:, 0].reshape(Hh, Ww),
                                   nz0[:, 1].reshape(Hh, Ww), scA.depth_min, scA.depth_max)
    d0[:, Ww // 2:] = scA.depths[ref][:, Ww // 2:]
    c0 = pm._compute_patch_cost(tv["ref_gray"], torch.from_numpy(d0), None, tv["src_grays"], tv["K"],
                                tv["K_inv"], tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"],
                                tv["x_grid"], tv["y_grid"]).numpy()

    # g04 spatial propagation (a10), forward (even iteration) and backward (odd)
    out = {}
    for tag, fwd in (("even", True), ("odd", False)):
        d, n, c = pm._spatial_propagation(torch.from_numpy(d0), torch.from_numpy(n0), torch.from_numpy(c0),
                                          tv["ref_gray"], tv["src_grays"], tv["K"], tv["K_inv"], tv["R_ref"],
                                          tv["t_ref"], tv["src_Rs"], tv["src_ts"], tv["x_grid"], tv["y_grid"],
                                          forward=fwd)
        out.update({f"depth_{tag}": d.numpy(), f"normal_{tag}": n.numpy(), f"cost_{tag}": c.numpy()})
    save("g04_propagate", ref=ref, srcs=srcs4, patch=7, depth=d0, normal=n0, cost=c0, **out)

    # g05 random refinement (a11): iterations 0 and 2, two samples each, injected noise
    out = {}
    pm2 = ref_pm(scA, 7, 3, 2)
    for it in (0, 2):
        inj = InjectedRng(SEED, ref, 1 + it * 2, with_init=False)
        with injected(inj):
            d, n, c = pm2._random_refinement(torch.from_numpy(d0), torch.from_numpy(n0), torch.from_numpy(c0),
                                             tv["ref_gray"], tv["src_grays"], tv["K"], tv["K_inv"],
                                             tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"],
                                             tv["x_grid"], tv["y_grid"], it)
        out.update({f"depth_it{it}": d.numpy(), f"normal_it{it}": n.numpy(), f"cost_it{it}": c.numpy()})
    save("g05_refine", ref=ref, srcs=srcs4, patch=7, samples=2, seed=SEED, depth=d0, normal=n0, cost=c0, **out)

    # ---- g06 end to end (a7): 96x128, 5 views, k=7, 3 iters x 4 samples ---------------------
    scB, g8B = quantised_scene(5, 96, 128, seed=12)
    save("scene_b", **scene_arrays(scB, g8B))
    pmB = ref_pm(scB, 7, 3, 4)
    posesB = ref_poses(scB)
    procB = {i: {"gray": scB.grays[i], "color": scB.colors[i], "shape": scB.grays[i].shape} for i in posesB}
    out = {}
    dmaps = {}
    for r in (0, 2):
        srcs = pmB._select_source_views(r, sorted(posesB), posesB, k=4)
        inj = InjectedRng(SEED, r, 0, with_init=True)
        with injected(inj):
            dm = pmB._patchmatch_cuda(r, srcs, procB, posesB)
        assert inj.draw == 1 + 3 * 4
        dmaps[r] = dm
        out.update({f"srcs_{r}": np.array(srcs), f"depth_{r}": dm.depth, f"normal_{r}": dm.normal,
                    f"confidence_{r}": dm.confidence})
    save("g06_patchmatch_e2e", refs=np.array([0, 2]), patch=7, iters=3, samples=4, seed=SEED, **out)

    # ---- g08 source selection (a6), g09 depth range (a4) -----------------------------------
    sc8 = make_scene(8, 8, 8, seed=1)
    poses8 = ref_poses(sc8)
    sel = np.array([pmB._select_source_views(r, sorted(poses8), poses8, k=4) for r in sorted(poses8)])
    # cameras with larger angular spread: some pairs fall outside the 5..60 degree window
    sc8b = make_scene(8, 8, 8, seed=1, arc_step_deg=17.0)
    poses8b = ref_poses(sc8b)
    selb = np.array([pmB._select_source_views(r, sorted(poses8b), poses8b, k=4) for r in sorted(poses8b)])
    save("g08_select_sources",
         R=np.stack([sc8.poses[i].R for i in range(8)]), t=np.stack([sc8.poses[i].t for i in range(8)]),
         selected=sel,
         Rb=np.stack([sc8b.poses[i].R for i in range(8)]), tb=np.stack([sc8b.poses[i].t for i in range(8)]),
         selected_b=selb)
    rng = np.random.default_rng(5)
    sparse = rng.normal(0, 0.8, (400, 3))
    pmr = ref_pm(scB, 7, 1, 1)
    pmr._estimate_depth_range(posesB, sparse)
    r1 = (pmr.depth_min, pmr.depth_max)
    pmr._estimate_depth_range(posesB, None)
    r2 = (pmr.depth_min, pmr.depth_max)
    save("g09_depth_range", sparse=sparse, with_sparse=np.array(r1), fallback=np.array(r2))

    # ---- g10 fusion + filter (a13, a14) -----------------------------------------------------
    gtmaps = {}
    for r in (0, 2, 4):
        conf = np.full(scB.depths[r].shape, 3.0, np.float32)
        conf[::3, ::2] = 2.0
        gtmaps[r] = mvs.DepthNormalMap(depth=scB.depths[r], normal=np.zeros(scB.depths[r].shape + (3,), np.float32),
                                       confidence=conf)
    pts, cols = pmB._fuse_depth_maps(gtmaps, procB, posesB)
    fpts, fcols = pmB._filter_points(pts, cols)
    save("g10_fuse_filter", refs=np.array([0, 2, 4]),
         confidence=np.stack([gtmaps[r].confidence for r in (0, 2, 4)]),
         points=pts, colors=cols, f_points=fpts, f_colors=fcols)

    # ---- g11 plane sweep (a15), g12 stereo back-projection / voxel / outliers (a17) ---------
    scC, g8C = quantised_scene(5, 48, 64, seed=13)
    save("scene_c", **scene_arrays(scC, g8C))
    dsC = stereo.DenseStereoReconstructor(refcam.Camera(K=scC.camera.K.copy(), dist=np.zeros(5)), scale=1.0,
                                          num_depths=16, patch_size=5)
    posesC = ref_poses(scC)
    procC = {i: {"gray": scC.grays[i], "color": scC.colors[i], "shape": scC.grays[i].shape} for i in posesC}
    nbrs = dsC._find_neighbors(2, sorted(posesC), posesC, k=6)
    inv = np.linspace(1 / scC.depth_max, 1 / scC.depth_min, 16)
    depths = 1.0 / inv
    dmap, conf, _ = dsC._plane_sweep_torch(scC.grays[2], scC.colors[2], posesC[2], nbrs, procC, posesC,
                                           depths, 48, 64)
    save("g11_plane_sweep", ref=2, nbrs=np.array(nbrs), depths=depths, patch=5, thresh=0.8,
         depth_map=dmap, confidence=conf)
    bp, bc = dsC._backproject(dmap, conf, scC.colors[2], posesC[2], min_confidence=dsC.min_views - 0.5)
    gp, gc = dsC._backproject(scC.depths[2], np.full((48, 64), 4.0, np.float32), scC.colors[2], posesC[2],
                              min_confidence=2.5)
    vp, vc = dsC._voxel_down_sample(gp, gc, voxel_size=0.02)
    op, oc = dsC._filter_outliers(gp, gc)
    save("g12_stereo_post", bp_points=bp, bp_colors=bc, gt_points=gp, gt_colors=gc, vox_points=vp, vox_colors=vc,
         out_points=op, out_colors=oc)

    # ---- g13 contracts ---------------------------------------------------------------------
    two = {0: posesB[0], 1: posesB[1]}
    p, c = pmB.reconstruct([{"image": scB.colors[0]}, {"image": scB.colors[1]}], two)
    p2, c2 = dsC.reconstruct([{"image": scC.colors[0]}, {"image": scC.colors[1]}], {0: posesC[0], 1: posesC[1]})
    save("g13_contracts", pm_points_shape=np.array(p.shape), pm_colors_shape=np.array(c.shape),
         st_points_shape=np.array(p2.shape), st_colors_shape=np.array(c2.shape))


if __name__ == "__main__":
    main()
