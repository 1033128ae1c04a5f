"""Bit-exact parity AT THE SIZES AND ON THE KERNEL VARIANTS bench.py times (BASELINE.json configs).

    config 3   16 x 1920x1080 PatchMatch, 7x7, S=4: the packed 8-bit sampling path, all 16 views in
               one launch, default strip height -- exactly the pm_step launch shape of the bench
    config 5   3840x2160 PatchMatch (one view of a 5-view scene)
    config 2   8 x 1280x720 plane sweep, 64 planes, 5x5, 6 neighbours, all 8 views in one launch

Each in both arithmetic modes, compared bit for bit with the matching mode of the CPU oracle on
the views checked (the oracle needs ~0.5 s per 1080p view for a 1-iteration x 2-sample schedule,
~3 s for the full 8 x (2 + 8) schedule the bench times -- test_config3_full_schedule_bit_exact).
Configs 4 and 5 as WORKLOADS (32 views sharded + gathered, 64 views of 4K + fusion):
tests/test_hip_workload_parity.py.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _eq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ " \
                       f"(first at {np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r})"


def _u8_scene(n, H, W, seed):
    """Synthetic scene rendered on the GPU, quantised to 8 bits as bench.py does (gray = code/255)."""
    import torch
    from amvs.synthetic import make_scene
    sc = make_scene(n, H, W, seed=seed, device="cuda" if torch.cuda.is_available() else "cpu")
    sc.grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
                for g in sc.grays]
    return sc


def _oracle_ctx(sc, ref, srcs, k, mode):
    from oracle import oracle
    oracle.set_threads(16)
    return oracle.ViewContext(sc.camera.K.astype(np.float32), sc.grays[ref], sc.poses[ref].R, sc.poses[ref].t,
                              [sc.grays[i] for i in srcs], [sc.poses[i].R for i in srcs],
                              [sc.poses[i].t for i in srcs], k, mode=mode)


@pytest.fixture(scope="module")
def scene_1080():
    return _u8_scene(16, 1080, 1920, 1234)


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_config3_16x1080p_batch_bit_exact(scene_1080, mode):
    import amvs
    from amvs.engine import make_pm_params
    sc = scene_1080
    H, W, n = 1080, 1920, 16
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        p = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max)
        depth, normal, conf = eng.patchmatch(ids, sources, p, 42)
        # the bench's launch shape: paired bands of 18 rows; fast arithmetic in groups of 4 views (two full
        # generations of waves), exact arithmetic as the whole batch
        assert (eng.last_views_per_launch(), eng.last_tile_rows()) == ((4, 18) if mode == "fast" else (16, 18))
    for r in (0, 9, 15):                                             # first, interior and last slot of the batch
        od, on, oc = _oracle_ctx(sc, r, sources[r], 7, mode).patchmatch(1, 2, sc.depth_min, sc.depth_max, 42, r)
        _eq(depth[r], od, f"{mode} 1080p view {r} depth")
        _eq(conf[r], oc, f"{mode} 1080p view {r} confidence")
        _eq(normal[r], on, f"{mode} 1080p view {r} normal")
    assert np.isfinite(depth).all()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_config3_full_schedule_bit_exact(scene_1080, mode):
    """The schedule bench.py times -- 8 iterations x (2 propagation + 8 refinement) hypotheses,
    mvs_patchmatch.py:287-308 -- on the bench's launch shape (16 views of 1920x1080 in one launch),
    two views of the batch against the oracle: the late-iteration regime (perturbation range
    depth_range / 128, coherent gathers, the lean-reciprocal range check, the LDS normal queue under
    every win rate from ~50 % down to ~1 %) at full size, not only on the 56x72 golden g17."""
    import amvs
    from amvs.engine import make_pm_params
    sc = scene_1080
    H, W, n = 1080, 1920, 16
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        p = make_pm_params(7, 8, 8, sc.depth_min, sc.depth_max)
        depth, normal, conf = eng.patchmatch(ids, sources, p, 42)
        assert eng.last_views_per_launch() == (4 if mode == "fast" else 16)
        assert eng.timing()["sweep_launches"] == (4 * 80 if mode == "fast" else 80)
    for r in (3, 12):
        od, on, oc = _oracle_ctx(sc, r, sources[r], 7, mode).patchmatch(8, 8, sc.depth_min, sc.depth_max, 42, r)
        _eq(depth[r], od, f"{mode} 1080p full schedule view {r} depth")
        _eq(conf[r], oc, f"{mode} 1080p full schedule view {r} confidence")
        _eq(normal[r], on, f"{mode} 1080p full schedule view {r} normal")
    assert np.isfinite(depth).all()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_config5_4k_view_bit_exact(mode):
    import amvs
    from amvs.engine import make_pm_params
    H, W, n = 2160, 3840, 5
    sc = _u8_scene(n, H, W, 4)
    ref, srcs = 2, [1, 3, 0, 4]
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in range(n):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        depth, normal, conf = eng.patchmatch([ref], [srcs], make_pm_params(7, 1, 1, sc.depth_min, sc.depth_max), 7)
    od, on, oc = _oracle_ctx(sc, ref, srcs, 7, mode).patchmatch(1, 1, sc.depth_min, sc.depth_max, 7, ref)
    _eq(depth[0], od, f"{mode} 4K depth")
    _eq(conf[0], oc, f"{mode} 4K confidence")
    _eq(normal[0], on, f"{mode} 4K normal")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_config5_4k_full_schedule_on_the_whole_batch_policy(mode):
    """Round 4: the late-iteration regime at 3840x2160.  Images wider than 2048 columns select
    another launch policy than 1080p -- the whole 16-view batch per launch (default_views_per_launch), strips of the
    reduced height pick_tile_rows chooses beyond 3072 columns -- and that policy had only met a 1 x (2 + 1) schedule.
    Here: 16 views of 4K, the full 8 x (2 + 8) schedule, one view of the batch against the oracle (~12 s)."""
    import amvs
    from amvs.engine import make_pm_params
    H, W, n = 2160, 3840, 16
    sc = _u8_scene(n, H, W, 1234)
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    import torch
    dev = torch.device("cuda", 0)
    hw = H * W
    d = torch.empty((n, hw), dtype=torch.float32, device=dev)
    nr = torch.empty((n, 3 * hw), dtype=torch.float32, device=dev)
    cf = torch.empty((n, hw), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        eng.patchmatch_device(ids, sources, make_pm_params(7, 8, 8, sc.depth_min, sc.depth_max), 42,
                              d.data_ptr(), nr.data_ptr(), cf.data_ptr())
        eng.sync()
        assert eng.last_views_per_launch() == 16 and eng.timing()["sweep_launches"] == 80
    r = 9
    od, on, oc = _oracle_ctx(sc, r, sources[r], 7, mode).patchmatch(8, 8, sc.depth_min, sc.depth_max, 42, r)
    _eq(d[r].cpu().numpy().reshape(H, W), od, f"{mode} 4K full schedule view {r} depth")
    _eq(cf[r].cpu().numpy().reshape(H, W), oc, f"{mode} 4K full schedule view {r} confidence")
    _eq(nr[r].cpu().numpy().reshape(H, W, 3), on, f"{mode} 4K full schedule view {r} normal")
    assert bool(torch.isfinite(d).all())


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_config2_8x720p_plane_sweep_bit_exact(mode):
    import torch

    import amvs
    H, W, n, D, k, S = 720, 1280, 8, 64, 5, 6
    sc = _u8_scene(n, H, W, 1234)
    ids = sorted(sc.poses)
    ds = amvs.DenseStereoReconstructor.__new__(amvs.DenseStereoReconstructor)
    nbrs = [ds._find_neighbors(r, ids, sc.poses, k=S) for r in ids]
    depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, D)).astype(np.float32)
    dev = torch.device("cuda", 0)
    dmap = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    conf = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        torch.cuda.synchronize()
        eng.plane_sweep_device(ids, nbrs, depths, k, 0.8, dmap.data_ptr(), conf.data_ptr())   # 8 views, one launch
        eng.sync()
        assert eng.last_tile_rows() == 60            # 12 bands of 60 rows: 8-bit running-best keys (chunks of 8 planes)
        # the 16-bit keys' shape as well (23 bands of 32 rows; what every launch ran before round 4)
        dmap32 = torch.empty_like(dmap)
        conf32 = torch.empty_like(conf)
        eng.set_sweep_tuning(tile_rows=32)
        torch.cuda.synchronize()
        eng.plane_sweep_device(ids, nbrs, depths, k, 0.8, dmap32.data_ptr(), conf32.data_ptr())
        eng.sync()
        assert eng.last_tile_rows() == 32
    assert torch.equal(dmap, dmap32) and torch.equal(conf, conf32)
    dmap, conf = dmap.cpu().numpy(), conf.cpu().numpy()
    for r in (0, 5):
        od, oc = _oracle_ctx(sc, r, nbrs[r], k, mode).plane_sweep(depths, 0.8)
        _eq(dmap[r], od, f"{mode} 720p sweep view {r} depth")
        _eq(conf[r], oc, f"{mode} 720p sweep view {r} confidence")
    assert conf.max() >= 3


def test_strip_height_follows_wave_quantisation(scene_1080):
    """pick_tile_rows (csrc/amvs_capi.hip): the automatic strip height makes the wave count of a launch
    land just below a whole number of generations of resident waves (256 CUs x 16 waves) under the
    locality cap of the paired-band schedule (3 k - 3 rows) -- 18 rows for 4 and for 2 views per launch,
    12 for one -- and the maps do not depend on it."""
    import amvs
    from amvs.engine import make_pm_params
    sc = scene_1080
    H, W = 1080, 1920
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    with amvs.Engine(H, W, 16, sc.camera.K.astype(np.float32), mode="fast") as eng:
        for i in ids:
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        got = {}
        maps = {}
        for n in (16, 4, 2, 1):
            p = make_pm_params(7, 1, 1, sc.depth_min, sc.depth_max, views_per_launch=n)
            maps[n] = eng.patchmatch(ids[:4], sources[:4], p, 3)
            got[n] = eng.last_tile_rows()
    assert got == {16: 18, 4: 18, 2: 18, 1: 12}, got          # (a 4-view batch caps views_per_launch at 4; paired bands)
    for n in (4, 2, 1):
        for a, b, what in zip(maps[16], maps[n], ("depth", "normal", "confidence")):
            _eq(b, a, f"{what}, {n} views per launch")
