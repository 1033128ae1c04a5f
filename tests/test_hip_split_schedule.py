"""The split schedule of the fast mode (amvs_pm_params.schedule = AMVS_SCHEDULE_SPLIT: a sampling kernel
without strip halo + a window kernel that streams its sample maps, pipelined over view groups on
several streams) returns the fused kernel's maps BIT FOR BIT -- same arithmetic, same order of every
sum -- and the fused kernel is the one pinned against the CPU oracle and the reference goldens
(test_hip_fast_parity.py, test_hip_fullsize_parity.py).  Shapes cover ragged widths (not multiples
of 64 or of the 58-column strips), uneven view groups, every compiled patch size and 2..6 sources;
the bench configuration (16 x 1080p, k = 7, S = 4) is compared at full size, and two views of it
against the oracle directly.
"""
import numpy as np
import pytest

from test_hip_fullsize_parity import _eq, _u8_scene, _oracle_ctx

pytestmark = pytest.mark.gpu


def _run(eng, refs, srcs, k, iters, samples, dmin, dmax, schedule, seed=7, vpl=0, tile_rows=0):
    from amvs.engine import make_pm_params
    p = make_pm_params(k, iters, samples, dmin, dmax, mode="fast", schedule=schedule, views_per_launch=vpl,
                       tile_rows=tile_rows)
    return eng.patchmatch(refs, srcs, p, seed)


def _engine(sc, mode="fast"):
    import amvs
    ids = sorted(sc.poses)
    H, W = sc.grays[ids[0]].shape
    eng = amvs.Engine(H, W, len(ids), sc.camera.K.astype(np.float32), mode=mode)
    for i in ids:
        eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
    return eng


@pytest.mark.parametrize("k,S,H,W,n_ref", [(7, 4, 97, 131, 5), (3, 2, 40, 64, 1), (5, 3, 70, 200, 2),
                                           (9, 5, 64, 129, 3), (11, 6, 83, 190, 7), (7, 4, 150, 58, 4),
                                           (13, 4, 60, 131, 3), (15, 3, 48, 104, 2), (19, 4, 50, 120, 2),
                                           (23, 6, 50, 120, 2)])
def test_split_equals_fused(k, S, H, W, n_ref):
    sc = _u8_scene(max(n_ref, S + 1), H, W, 100 + k)
    ids = sorted(sc.poses)
    refs = ids[:n_ref]
    srcs = [[j for j in ids if j != r][:S] for r in refs]
    with _engine(sc) as eng:
        fused = _run(eng, refs, srcs, k, 3, 2, sc.depth_min, sc.depth_max, "view-major")
        for groups, rows, lds in ((0, 0, 0), (3, 5, 16384), (8, 40, 40960)):
            eng.set_split_tuning(groups, rows, lds)
            split = _run(eng, refs, srcs, k, 3, 2, sc.depth_min, sc.depth_max, "split")
            for a, b, what in zip(fused, split, ("depth", "normal", "confidence")):
                _eq(b, a, f"{what} (groups={groups}, sample rows={rows})")


def test_split_needs_fast_mode():
    import amvs
    from amvs.engine import make_pm_params
    sc = _u8_scene(5, 40, 64, 3)
    with _engine(sc, mode="exact") as eng:
        p = make_pm_params(7, 1, 1, sc.depth_min, sc.depth_max, mode="exact", schedule="split")
        with pytest.raises(amvs.AmvsError, match="fast mode only"):
            eng.patchmatch([0], [[1, 2, 3, 4]], p, 1)
    with _engine(sc, mode="fast") as eng:               # ... and a patch size the kernels are compiled for
        p = make_pm_params(31, 1, 1, sc.depth_min, sc.depth_max, mode="fast", schedule="split")
        with pytest.raises(amvs.AmvsError, match="compiled patch sizes"):
            eng.patchmatch([0], [[1, 2, 3, 4]], p, 1)


def test_split_config3_16x1080p_equals_fused_and_oracle():
    import amvs
    sc = _u8_scene(16, 1080, 1920, 1234)
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    with _engine(sc) as eng:
        fused = _run(eng, ids, sources, 7, 1, 2, sc.depth_min, sc.depth_max, "view-major", seed=5)
        split = _run(eng, ids, sources, 7, 1, 2, sc.depth_min, sc.depth_max, "split", seed=5)
        assert eng.last_views_per_launch() == 16
    for a, b, what in zip(fused, split, ("depth", "normal", "confidence")):
        _eq(b, a, what)
    for ref in (0, 9):
        ctx = _oracle_ctx(sc, ref, sources[ref], 7, "fast")
        d, nrm, conf = ctx.patchmatch(1, 2, sc.depth_min, sc.depth_max, 5, ref)
        _eq(split[0][ref], d, f"depth of view {ref} vs oracle")
        _eq(split[2][ref], conf, f"confidence of view {ref} vs oracle")
