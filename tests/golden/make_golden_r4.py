"""Round-4 golden vectors captured from the reference (build container only).

    python tests/golden/make_golden_r4.py

Same rules as make_golden.py / make_golden_r2.py (whose fixtures stay byte-identical): the reference is
imported from /root/reference through ref_loader.py, driven on the committed small scenes with this
repository's counter-hash RNG injected in place of torch.rand / torch.randn, and only inputs + the
reference's outputs are stored.  The reference takes ANY patch_size (mvs_patchmatch.py:45, :396-397;
dense_stereo.py:36, :325-341); these fixtures pin the patch sizes that run on the run-time-k kernels
(csrc/amvs_generic.hip):

    g19_patch_cost_k13_15   _compute_patch_cost for patch sizes 13 and 15 (scene A, S = 4, g03's depth map)
    g20_patchmatch_k13      _patchmatch_cuda with patch 13, 3 iterations x 4 samples, one view of scene D
    g21_plane_sweep_k13     _plane_sweep_torch with patch 13, 6 neighbours, 16 planes (scene D)

and the operating points of the reference's CLI (run_reconstruction.py:131-136, :150-154), which the classes'
defaults reproduce:

    g22_patchmatch_cli      _patchmatch_cuda with the constructor defaults the CLI leaves alone -- patch 11, 8 samples --
                            and its num_iterations = 3, two views of scene D
    g23_plane_sweep_cli     _plane_sweep_torch with DenseStereoReconstructor's defaults: 64 planes, patch 5, 6
                            neighbours, threshold 0.8 (scene D)
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


def main():
    torch.set_num_threads(4)

    # ---- g19: patch cost for k = 13, 15 (scene A as g03 / g15) -----------------------------------
    scA, _ = mg.quantised_scene(5, 64, 96, seed=11)
    g03 = np.load(os.path.join(HERE, "g03_patch_cost.npz"))
    ref, srcs4, depth = int(g03["ref"]), list(g03["srcs4"]), g03["depth"]
    out = {}
    for k in (13, 15):
        pm = mg.ref_pm(scA, k, 1, 1)
        tv = mg.torch_view(pm, scA, ref, srcs4)
        out[f"cost_k{k}"] = pm._compute_patch_cost(
            tv["ref_gray"], torch.from_numpy(depth), None, tv["src_grays"], tv["K"], tv["K_inv"],
            tv["R_ref"], tv["t_ref"], tv["src_Rs"], tv["src_ts"], tv["x_grid"], tv["y_grid"]).numpy()
    mg.save("g19_patch_cost_k13_15", ref=ref, srcs=np.array(srcs4), depth=depth, **out)

    # ---- scene D (the committed one: 7 views 56x72) ----------------------------------------------
    scD, _ = mg.quantised_scene(7, 56, 72, seed=14)
    posesD = mg.ref_poses(scD)
    procD = {i: {"gray": scD.grays[i], "color": scD.colors[i], "shape": scD.grays[i].shape} for i in posesD}

    # ---- g20: PatchMatch end to end with a 13x13 patch -------------------------------------------
    pmD = mg.ref_pm(scD, 13, 3, 4)
    r = 3
    srcs = pmD._select_source_views(r, sorted(posesD), posesD, k=4)
    inj = mg.InjectedRng(SEED, r, 0, with_init=True)
    with mg.injected(inj):
        dm = pmD._patchmatch_cuda(r, srcs, procD, posesD)
    assert inj.draw == 1 + 3 * 4
    mg.save("g20_patchmatch_k13", ref=r, srcs=np.array(srcs), patch=13, iters=3, samples=4, seed=SEED,
            depth=dm.depth, normal=dm.normal, confidence=dm.confidence)

    # ---- g21: plane sweep with a 13x13 patch, 6 neighbours ---------------------------------------
    dsD = stereo.DenseStereoReconstructor(refcam.Camera(K=scD.camera.K.copy(), dist=np.zeros(5)), scale=1.0,
                                          num_depths=16, patch_size=13)
    nbrs = dsD._find_neighbors(3, sorted(posesD), posesD, k=6)
    depths = 1.0 / np.linspace(1 / scD.depth_max, 1 / scD.depth_min, 16)
    dmap, conf, _ = dsD._plane_sweep_torch(scD.grays[3], scD.colors[3], posesD[3], nbrs, procD, posesD,
                                           depths, 56, 72)
    mg.save("g21_plane_sweep_k13", ref=3, nbrs=np.array(nbrs), depths=depths, patch=13, thresh=0.8,
            depth_map=dmap, confidence=conf)

    # ---- g22: the CLI's PatchMatch operating point (patch 11, 3 iterations x 8 samples) ----------
    pmC = mg.ref_pm(scD, 11, 3, 8)
    out = {}
    for r in (1, 4):
        srcs = pmC._select_source_views(r, sorted(posesD), posesD, k=4)
        inj = mg.InjectedRng(SEED, r, 0, with_init=True)
        with mg.injected(inj):
            dm = pmC._patchmatch_cuda(r, srcs, procD, posesD)
        assert inj.draw == 1 + 3 * 8
        out.update({f"srcs_{r}": np.array(srcs), f"depth_{r}": dm.depth, f"normal_{r}": dm.normal,
                    f"confidence_{r}": dm.confidence})
    mg.save("g22_patchmatch_cli", refs=np.array([1, 4]), patch=11, iters=3, samples=8, seed=SEED, **out)

    # ---- g23: the stereo class's defaults (64 planes, patch 5, 6 neighbours) ----------------------
    dsC = stereo.DenseStereoReconstructor(refcam.Camera(K=scD.camera.K.copy(), dist=np.zeros(5)), scale=1.0)
    assert (dsC.num_depths, dsC.patch_size, dsC.consistency_thresh) == (64, 5, 0.8)
    nbrs = dsC._find_neighbors(2, sorted(posesD), posesD, k=6)
    depths = 1.0 / np.linspace(1 / scD.depth_max, 1 / scD.depth_min, dsC.num_depths)
    dmap, conf, _ = dsC._plane_sweep_torch(scD.grays[2], scD.colors[2], posesD[2], nbrs, procD, posesD,
                                           depths, 56, 72)
    mg.save("g23_plane_sweep_cli", ref=2, nbrs=np.array(nbrs), depths=depths, patch=5, thresh=0.8,
            depth_map=dmap, confidence=conf)


if __name__ == "__main__":
    main()
