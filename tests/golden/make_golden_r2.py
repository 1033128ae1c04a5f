"""Round-2 golden vectors captured from the reference (build container only).

    python tests/golden/make_golden_r2.py

Same rules as make_golden.py (whose fixtures g01-g13 stay byte-identical): the reference is
imported from /root/reference through ref_loader.py, driven on small synthetic scenes with this
repository's counter-hash RNG injected in place of torch.rand / torch.randn, and only inputs +
the reference's outputs are stored.

    g14_fused_cloud      the reference's _fuse_depth_maps + _filter_points (mvs_patchmatch.py:536-588)
                         applied to ITS OWN g06 depth maps: the cloud the Chamfer acceptance test
                         compares the HIP cloud with
    scene_d              7 views 56x72 (plane sweep with 6 neighbours needs 7 views)
    g15_patch_cost_k359  _compute_patch_cost for patch sizes 3, 5, 9 (scene A, S=4)
    g16_plane_sweep_s6   _plane_sweep_torch with 6 neighbours, 24 planes (scene D)
    g17_patchmatch_long  _patchmatch_cuda with the BASELINE schedule (8 iterations x 8 samples,
                         7x7, S=4) on one view of scene D: long enough for chaotic divergence
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (imports the reference through ref_loader)

mvs, stereo, refcam = mg.mvs, mg.stereo, mg.refcam
SEED = mg.SEED


def golden_scene(name):
    """Rebuild a Scene-like object from a committed scene_*.npz (so g14 fuses exactly g06's scene)."""
    g = np.load(os.path.join(HERE, name + ".npz"))

    class S:
        pass
    sc = S()
    sc.grays = [x.astype(np.float32) / np.float32(255.0) for x in g["gray_u8"]]
    sc.colors = list(g["color_u8"])
    sc.K = g["K"]
    sc.poses = {i: refcam.CameraPose(R=g["R"][i].copy(), t=g["t"][i].copy()) for i in range(len(sc.grays))}
    sc.depth_min, sc.depth_max = float(g["depth_min"]), float(g["depth_max"])
    return sc


def main():
    torch.set_num_threads(4)

    # ---- g14: fused cloud of the reference's own g06 maps ----------------------------------
    scB = golden_scene("scene_b")
    g06 = np.load(os.path.join(HERE, "g06_patchmatch_e2e.npz"))
    out = {}
    for min_views in (2, 3):
        pm = mvs.PatchMatchMVS(refcam.Camera(K=scB.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                               num_iterations=3, num_samples=4, min_views=min_views)
        proc = {i: {"gray": scB.grays[i], "color": scB.colors[i], "shape": scB.grays[i].shape} for i in scB.poses}
        maps = {int(r): mvs.DepthNormalMap(depth=g06[f"depth_{int(r)}"], normal=g06[f"normal_{int(r)}"],
                                           confidence=g06[f"confidence_{int(r)}"]) for r in g06["refs"]}
        pts, cols = pm._fuse_depth_maps(maps, proc, scB.poses)
        fpts, fcols = pm._filter_points(pts, cols)
        out.update({f"points_mv{min_views}": pts, f"colors_mv{min_views}": cols,
                    f"f_points_mv{min_views}": fpts, f"f_colors_mv{min_views}": fcols})
        print(f"g14 min_views={min_views}: {len(pts)} fused, {len(fpts)} after the filter")
    mg.save("g14_fused_cloud", refs=g06["refs"], **out)

    # ---- g15: patch cost for k = 3, 5, 9 (scene A as g03) -----------------------------------
    scA, _ = mg.quantised_scene(5, 64, 96, seed=11)
    g03 = np.load(os.path.join(HERE, "g03_patch_cost.npz"))
    ref, srcs4, depth = int(g03["ref"]), list(g03["srcs4"]), g03["depth"]
    out = {}
    for k in (3, 5, 9):
        pm = mg.ref_pm(scA, k, 1, 1)
        tv = mg.torch_view(pm, scA, ref, srcs4)
        out[f"cost_k{k}"] = pm._compute_patch_cost(
            tv["ref_gray"], torch.from_numpy(depth), None, tv["src_grays"], tv["K"], tv["K_inv"],
            tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"], tv["x_grid"], tv["y_grid"]).numpy()
    mg.save("g15_patch_cost_k359", ref=ref, srcs=np.array(srcs4), depth=depth, **out)

    # ---- scene D: 7 views 56x72 -----------------------------------------------------------------
    scD, g8D = mg.quantised_scene(7, 56, 72, seed=14)
    mg.save("scene_d", **mg.scene_arrays(scD, g8D))
    posesD = mg.ref_poses(scD)
    procD = {i: {"gray": scD.grays[i], "color": scD.colors[i], "shape": scD.grays[i].shape} for i in posesD}

    # ---- g16: plane sweep with 6 neighbours -------------------------------------------------
    dsD = stereo.DenseStereoReconstructor(refcam.Camera(K=scD.camera.K.copy(), dist=np.zeros(5)), scale=1.0,
                                          num_depths=24, patch_size=5)
    nbrs = dsD._find_neighbors(3, sorted(posesD), posesD, k=6)
    assert len(nbrs) == 6
    depths = 1.0 / np.linspace(1 / scD.depth_max, 1 / scD.depth_min, 24)
    dmap, conf, _ = dsD._plane_sweep_torch(scD.grays[3], scD.colors[3], posesD[3], nbrs, procD, posesD,
                                           depths, 56, 72)
    mg.save("g16_plane_sweep_s6", ref=3, nbrs=np.array(nbrs), depths=depths, patch=5, thresh=0.8,
            depth_map=dmap, confidence=conf)

    # ---- g17: the BASELINE schedule end to end ----------------------------------------------
    pmD = mg.ref_pm(scD, 7, 8, 8)
    r = 3
    srcs = pmD._select_source_views(r, sorted(posesD), posesD, k=4)
    inj = mg.InjectedRng(SEED, r, 0, with_init=True)
    with mg.injected(inj):
        dm = pmD._patchmatch_cuda(r, srcs, procD, posesD)
    assert inj.draw == 1 + 8 * 8
    mg.save("g17_patchmatch_long", ref=r, srcs=np.array(srcs), patch=7, iters=8, samples=8, seed=SEED,
            depth=dm.depth, normal=dm.normal, confidence=dm.confidence)


if __name__ == "__main__":
    main()
