"""Any odd patch size (round 4) against the CPU oracle, whose k is a run-time argument and which
tests/test_oracle_modes_golden.py pins at k = 13, 15 against the reference's own outputs (g19, g20, g21).  The
reference takes any patch_size (mvs_patchmatch.py:45, :396-397; dense_stereo.py:36, :325-341); the compiled kernels
cover 3 ... 29 (13 ... 29 since the end of round 4: two to four times the run-time-k kernels' rate); 31, the largest
size, runs in the run-time-k kernels of csrc/amvs_generic.hip -- every test here visits both kinds.

Bar: BIT-EXACT against the oracle in both arithmetic modes; the reference tolerances of tests/conftest.py against
the goldens.
"""
import numpy as np
import pytest

from conftest import CONF_HIST_TOL, E2E_MIN_FRACTION, assert_cost_close, load_golden

pytestmark = pytest.mark.gpu

MODES = ("exact", "fast")


def _eq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ " \
                       f"(first at {np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r})"


@pytest.fixture(scope="module", params=MODES)
def eng_mode(request, scene_a):
    eng = scene_a.engine(request.param)
    yield eng, request.param
    eng.close()


def _mixed_depth(scene, ref, seed):
    rng = np.random.default_rng(seed)
    d = np.exp(rng.uniform(np.log(scene.depth_min), np.log(scene.depth_max), (scene.H, scene.W))).astype(np.float32)
    d[:, scene.W // 2:] = scene.gt_depth[ref][:, scene.W // 2:]
    d[:5, :7] = np.float32(0.05)
    d[-6:, -9:] = np.float32(400.0)
    return d


@pytest.mark.parametrize("k", [13, 15, 21, 31])
def test_box_stats_bit_exact(scene_a, k):
    from oracle import oracle
    with scene_a.engine() as eng:
        for v in (0, 3):
            m, var = eng.box_stats(v, k)
            om, ovar = oracle.box_stats(scene_a.grays[v], k)
            _eq(m, om, f"mean k{k}")
            _eq(var, ovar, f"var k{k}")


@pytest.mark.parametrize("k", [13, 15, 17, 19, 23, 31])
@pytest.mark.parametrize("srcs", [[1, 3, 0, 4], [3, 1], [0, 1, 4]])
def test_eval_cost_bit_exact(eng_mode, scene_a, k, srcs):
    eng, mode = eng_mode
    ref = 2
    depth = _mixed_depth(scene_a, ref, 5)
    got = eng.eval_cost(ref, srcs, k, depth)
    want = scene_a.oracle_ctx(ref, srcs, k, mode).patch_cost(depth)
    assert np.isfinite(want).any()
    _eq(got, want, f"{mode} cost k{k} S{len(srcs)}")


def test_eval_cost_vs_reference_golden(eng_mode):
    eng, mode = eng_mode
    g = load_golden("g19_patch_cost_k13_15")
    for k in (13, 15):
        got = eng.eval_cost(int(g["ref"]), list(g["srcs"]), k, g["depth"])
        if mode == "exact":
            assert_cost_close(got, g[f"cost_k{k}"], 1e-4, f"k{k}")
        else:
            fin = np.isfinite(got) & np.isfinite(g[f"cost_k{k}"])
            assert (np.isfinite(got) != np.isfinite(g[f"cost_k{k}"])).sum() <= 3
            assert np.quantile(np.abs(got - g[f"cost_k{k}"])[fin], 0.999) < 1e-4


def test_confidence_bit_exact(eng_mode, scene_a):
    eng, mode = eng_mode
    g = load_golden("g07_confidence")
    ref, srcs = int(g["ref"]), list(g["srcs"])
    for k in (13, 19, 21, 31):
        got = eng.confidence(ref, srcs, k, g["depth"])
        _eq(got, scene_a.oracle_ctx(ref, srcs, k, mode).confidence(g["depth"]), f"{mode} confidence k{k}")


@pytest.mark.parametrize("k", [13, 17, 21, 27, 31])
@pytest.mark.parametrize("off", [(1, 0), (0, 1), (-1, 0), (0, -1)])
def test_propagate_step_bit_exact(eng_mode, scene_a, off, k):
    eng, mode = eng_mode
    g = load_golden("g04_propagate")
    ref, srcs = int(g["ref"]), list(g["srcs"])
    # (g04's cost map belongs to another patch size: a first evaluation gives this patch's costs)
    cost = scene_a.oracle_ctx(ref, srcs, k, mode).patch_cost(g["depth"])
    got = eng.propagate_step(ref, srcs, k, g["depth"], g["normal"], cost, off[0], off[1], scene_a.depth_min)
    want = scene_a.oracle_ctx(ref, srcs, k, mode).propagate_step(g["depth"], g["normal"], cost, off[0], off[1],
                                                                 scene_a.depth_min)
    for a, b, name in zip(got, want, ("depth", "normal", "cost")):
        _eq(a, b, f"{mode} propagate {off} {name}")
    assert (got[0] != g["depth"]).mean() > 0.01


@pytest.mark.parametrize("k", [15, 19, 23, 29, 31])
@pytest.mark.parametrize("it", [0, 2])
def test_refine_step_bit_exact(eng_mode, scene_a, it, k):
    from oracle import oracle
    eng, mode = eng_mode
    g = load_golden("g05_refine")
    ref, srcs, seed = int(g["ref"]), list(g["srcs"]), int(g["seed"])
    dr = np.float32((scene_a.depth_max - scene_a.depth_min) * 0.5 ** it)
    nr = np.float32(0.5 * 0.5 ** it)
    ctx = scene_a.oracle_ctx(ref, srcs, k, mode)
    d, n = g["depth"], g["normal"]
    c = ctx.patch_cost(d)
    od, on, oc = d, n, c
    for s in range(2):
        draw = 1 + it * 2 + s
        d, n, c = eng.refine_step(ref, srcs, k, d, n, c, seed, ref, draw, dr, nr, scene_a.depth_min, scene_a.depth_max)
        u, nz = oracle.rng_fill(seed, ref, draw, scene_a.H * scene_a.W)
        od, on, oc = ctx.refine_step(od, on, oc, u, nz, dr, nr, scene_a.depth_min, scene_a.depth_max)
        _eq(d, od, f"{mode} refine it{it} s{s} depth")
        _eq(c, oc, f"{mode} refine it{it} s{s} cost")
        _eq(n, on, f"{mode} refine it{it} s{s} normal")
    assert (d != g["depth"]).mean() > 0.005


@pytest.mark.parametrize("mode", MODES)
def test_patchmatch_k13_bit_exact_and_reference_golden(scene_d, mode):
    """_patchmatch_cuda with a 13x13 patch (g20): bit-exact against the oracle, within the reference tolerances
    against the reference's own maps; two views in one batch, several strips per view (tile_rows)."""
    from amvs.engine import make_pm_params
    g = load_golden("g20_patchmatch_k13")
    r, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    with scene_d.engine(mode) as eng:
        p = make_pm_params(k, int(g["iters"]), int(g["samples"]), scene_d.depth_min, scene_d.depth_max)
        depth, normal, conf = eng.patchmatch([r, 1], [srcs, [0, 2, 3, 4]], p, int(g["seed"]))
        assert eng.timing()["sweep_launches"] == 3 * (2 + 4)
        p2 = make_pm_params(k, int(g["iters"]), int(g["samples"]), scene_d.depth_min, scene_d.depth_max, tile_rows=9)
        depth2, normal2, conf2 = eng.patchmatch([r], [srcs], p2, int(g["seed"]))
    for i, (view, ss) in enumerate(((r, srcs), (1, [0, 2, 3, 4]))):
        od, on, oc = scene_d.oracle_ctx(view, ss, k, mode).patchmatch(int(g["iters"]), int(g["samples"]), scene_d.depth_min,
                                                                      scene_d.depth_max, int(g["seed"]), view)
        _eq(depth[i], od, f"{mode} view {view} depth")
        _eq(conf[i], oc, f"{mode} view {view} confidence")
        _eq(normal[i], on, f"{mode} view {view} normal")
    _eq(depth2[0], depth[0], f"{mode} 9-row strips depth")
    _eq(normal2[0], normal[0], f"{mode} 9-row strips normal")
    _eq(conf2[0], conf[0], f"{mode} 9-row strips confidence")
    rel = np.abs(depth[0] - g["depth"]) / g["depth"]
    assert np.mean(rel <= 1e-3) >= E2E_MIN_FRACTION
    hist_got = np.bincount(conf[0].astype(int).ravel(), minlength=5) / conf[0].size
    hist_ref = np.bincount(g["confidence"].astype(int).ravel(), minlength=5) / conf[0].size
    assert np.abs(hist_got - hist_ref).max() < CONF_HIST_TOL


@pytest.mark.parametrize("mode", MODES)
def test_plane_sweep_k13_bit_exact_and_reference_golden(scene_d, mode):
    g = load_golden("g21_plane_sweep_k13")
    ref, nbrs, k = int(g["ref"]), list(g["nbrs"]), int(g["patch"])
    depths = g["depths"].astype(np.float32)
    with scene_d.engine(mode) as eng:
        d, conf = eng.plane_sweep(ref, nbrs, depths, k, float(g["thresh"]))
        eng.set_sweep_tuning(tile_rows=5, planes_per_wave=3)              # several strips, several plane chunks
        d2, conf2 = eng.plane_sweep(ref, nbrs, depths, k, float(g["thresh"]))
        d3, conf3 = eng.plane_sweep(ref, nbrs[:3], depths, 31, -0.2)       # the division form of the fast vote (run-time-k kernel)
        d4, conf4 = eng.plane_sweep(ref, nbrs, depths, 25, 0.6)            # six sources at 25 x 25: a 128-bit validity history
    od, oc = scene_d.oracle_ctx(ref, nbrs, k, mode).plane_sweep(depths, float(g["thresh"]))
    _eq(d, od, f"{mode} depth")
    _eq(conf, oc, f"{mode} confidence")
    _eq(d2, od, f"{mode} depth (chunked)")
    _eq(conf2, oc, f"{mode} confidence (chunked)")
    od3, oc3 = scene_d.oracle_ctx(ref, nbrs[:3], 31, mode).plane_sweep(depths, -0.2)
    _eq(d3, od3, f"{mode} k31 depth")
    _eq(conf3, oc3, f"{mode} k31 confidence")
    od4, oc4 = scene_d.oracle_ctx(ref, nbrs, 25, mode).plane_sweep(depths, 0.6)
    _eq(d4, od4, f"{mode} k25 S6 depth")
    _eq(conf4, oc4, f"{mode} k25 S6 confidence")
    assert np.mean(conf == g["confidence"]) > 0.995 and np.mean(d == g["depth_map"]) > 0.99


@pytest.mark.parametrize("shape,k,S", [((33, 59), 13, 3), ((70, 117), 15, 3), ((35, 61), 17, 3), ((37, 66), 21, 3), ((41, 200), 25, 2), ((20, 64), 31, 3)])
def test_ragged_shapes_and_float_images_bit_exact(shape, k, S):
    """Widths that are not multiples of the strip's output width (64 - 2 (k/2)), heights below the patch size, and
    rendered float images (not 8-bit exact: the exact arithmetic samples the float32 maps, U8 = false)."""
    import amvs
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    H, W = shape
    sc = make_scene(4, H, W, seed=H)
    K = sc.camera.K.astype(np.float32)
    others = [0, 2, 3][:S]
    for quantise in (False, True):
        grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0) if quantise else g
                 for g in sc.grays]
        with amvs.Engine(H, W, 4, K) as eng:
            for i in range(4):
                eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
            assert eng.sampling_mode() == ("u8-pairs" if quantise else "f32")
            p = make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max)
            depth, normal, conf = eng.patchmatch([1], [others], p, 9)
        ctx = oracle.ViewContext(K, grays[1], sc.poses[1].R, sc.poses[1].t, [grays[i] for i in others],
                                 [sc.poses[i].R for i in others], [sc.poses[i].t for i in others], k)
        od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 9, 1)
        _eq(depth[0], od, f"{shape} k{k} depth (8-bit {quantise})")
        _eq(conf[0], oc, f"{shape} k{k} conf (8-bit {quantise})")
        _eq(normal[0], on, f"{shape} k{k} normal (8-bit {quantise})")


@pytest.mark.parametrize("k", [23, 29, 31])
@pytest.mark.parametrize("mode", MODES)
def test_largest_patch_with_six_sources(scene_d, mode, k):
    """k = 31 with S = 6: the largest LDS footprint of the run-time-k kernels (61.4 KB per wave for the sweep step,
    63.3 KB for the plane sweep -- just under the 64 KB a workgroup may take); k = 23 / 29 with S = 6: the compiled
    kernels with the most register-resident rings and a validity history of 72 / 90 bits (a 128-bit integer): cost
    evaluation, a short sweep and the plane sweep against the oracle."""
    from amvs.engine import make_pm_params
    ref, srcs = 3, [0, 1, 2, 4, 5, 6]
    depth = _mixed_depth(scene_d, ref, 3)
    depths = (1.0 / np.linspace(1 / scene_d.depth_max, 1 / scene_d.depth_min, 6)).astype(np.float32)
    with scene_d.engine(mode) as eng:
        cost = eng.eval_cost(ref, srcs, k, depth)
        d, n, c = eng.patchmatch([ref], [srcs], make_pm_params(k, 1, 2, scene_d.depth_min, scene_d.depth_max), 4)
        sd, sc_ = eng.plane_sweep(ref, srcs, depths, k, 0.5)
    ctx = scene_d.oracle_ctx(ref, srcs, k, mode)
    _eq(cost, ctx.patch_cost(depth), f"{mode} k{k} S6 cost")
    od, on, oc = ctx.patchmatch(1, 2, scene_d.depth_min, scene_d.depth_max, 4, ref)
    _eq(d[0], od, f"{mode} k{k} S6 depth")
    _eq(n[0], on, f"{mode} k{k} S6 normal")
    _eq(c[0], oc, f"{mode} k{k} S6 confidence")
    osd, osc = ctx.plane_sweep(depths, 0.5)
    _eq(sd, osd, f"{mode} k{k} S6 sweep depth")
    _eq(sc_, osc, f"{mode} k{k} S6 sweep confidence")


@pytest.mark.parametrize("patch", [13, 21, 31])
def test_classes_accept_any_odd_patch_size(scene_b, capsys, patch):
    """PatchMatchMVS(patch_size=13 / 21 / 31) / DenseStereoReconstructor(patch_size=13 / 21 / 31) run end to end (the
    reference's constructors take any patch size); even and oversized patches are refused with a message."""
    import amvs
    from amvs._lib import AmvsError
    cam = amvs.Camera(K=scene_b.K.copy(), dist=np.zeros(5))
    images = [{"image": c} for c in scene_b.colors]
    pm = amvs.PatchMatchMVS(cam, scale=1.0, patch_size=patch, num_iterations=2, num_samples=2, min_views=2,
                            depth_min=scene_b.depth_min, depth_max=scene_b.depth_max, seed=3)
    pm._estimate_depth_range = lambda poses, pts=None: None           # keep the scene's own range
    pts, cols = pm.reconstruct(images, scene_b.poses())
    assert pts.ndim == 2 and pts.shape[1] == 3 and len(pts) == len(cols)
    ds = amvs.DenseStereoReconstructor(cam, scale=1.0, num_depths=8, patch_size=patch, min_views=2)
    pts2, cols2 = ds.reconstruct(images, scene_b.poses())
    assert pts2.ndim == 2 and len(pts2) == len(cols2)
    capsys.readouterr()
    with scene_b.engine() as eng:
        for bad in (6, 33, 1):
            with pytest.raises(AmvsError, match="patch_size"):
                eng.eval_cost(1, [0, 2], bad, scene_b.gt_depth[1])


@pytest.mark.parametrize("mode", MODES)
def test_cli_operating_points_bit_exact_and_reference_golden(scene_d, mode):
    """The reference CLI's operating points (the classes' defaults): _patchmatch_cuda with patch 11, 3 iterations x
    8 samples (g22) and _plane_sweep_torch with 64 planes, patch 5, 6 neighbours (g23) -- the HIP kernels bit-exact
    against the oracle, and against the reference's own maps within the stated tolerances (measured: every depth
    within 1e-3, 96 % identical, confidence identical)."""
    from amvs.engine import make_pm_params
    g = load_golden("g22_patchmatch_cli")
    refs = [int(x) for x in g["refs"]]
    srcs = [list(g[f"srcs_{r}"]) for r in refs]
    k, iters, samples, seed = int(g["patch"]), int(g["iters"]), int(g["samples"]), int(g["seed"])
    s23 = load_golden("g23_plane_sweep_cli")
    planes = s23["depths"].astype(np.float32)
    with scene_d.engine(mode) as eng:
        depth, normal, conf = eng.patchmatch(refs, srcs, make_pm_params(k, iters, samples, scene_d.depth_min, scene_d.depth_max), seed)
        sd, sc = eng.plane_sweep(int(s23["ref"]), list(s23["nbrs"]), planes, int(s23["patch"]), float(s23["thresh"]))
    for i, r in enumerate(refs):
        od, on, oc = scene_d.oracle_ctx(r, srcs[i], k, mode).patchmatch(iters, samples, scene_d.depth_min, scene_d.depth_max, seed, r)
        _eq(depth[i], od, f"{mode} view {r} depth")
        _eq(conf[i], oc, f"{mode} view {r} confidence")
        _eq(normal[i], on, f"{mode} view {r} normal")
        rel = np.abs(depth[i] - g[f"depth_{r}"]) / g[f"depth_{r}"]
        assert np.mean(rel <= 1e-3) >= E2E_MIN_FRACTION
        hist_got = np.bincount(conf[i].astype(int).ravel(), minlength=5) / conf[i].size
        hist_ref = np.bincount(g[f"confidence_{r}"].astype(int).ravel(), minlength=5) / conf[i].size
        assert np.abs(hist_got - hist_ref).max() < CONF_HIST_TOL
    od, oc = scene_d.oracle_ctx(int(s23["ref"]), list(s23["nbrs"]), int(s23["patch"]), mode).plane_sweep(planes, float(s23["thresh"]))
    _eq(sd, od, f"{mode} sweep depth")
    _eq(sc, oc, f"{mode} sweep confidence")
    assert np.mean(sc == s23["confidence"]) > 0.995 and np.mean(sd == s23["depth_map"]) > 0.99
