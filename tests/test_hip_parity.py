"""HIP kernels (through the C ABI) against the CPU oracle and the reference's golden vectors.

Bar: BIT-EXACT against the oracle for depth, cost, confidence and normals (the kernels and the
oracle use the same arithmetic order by construction); against the golden vectors captured
from the reference the tolerances of tests/test_oracle_golden.py apply (box-filter summation
order inside oneDNN is unobservable).
"""
import numpy as np
import pytest

from conftest import CONF_HIST_TOL, E2E_MIN_FRACTION, GoldenScene, assert_cost_close, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amvs_mod():
    import amvs
    return amvs


@pytest.fixture(scope="module")
def eng_a(scene_a, amvs_mod):
    eng = scene_a.engine()
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def eng_b(scene_b, amvs_mod):
    eng = scene_b.engine()
    yield eng
    eng.close()


def _eq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ " \
                       f"(first at {np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r})"


# ------------------------------------------------------------------ primitives ----
def test_lean_math_exhaustive(eng_a):
    """rcp_rn / sqrt_rn (v_rcp / v_rsq + FMA corrections, IEEE path out of range) equal 1.0f/x and
    sqrtf(x) on every one of the 2^32 float bit patterns: the replacement of the IEEE expansions
    cannot change a single bit anywhere."""
    bad_rcp, bad_sqrt = eng_a.selftest_lean_math()
    assert (bad_rcp, bad_sqrt) == (0, 0)



def test_rng_bit_exact(eng_a):
    from oracle import oracle
    for seed, view, draw, n in ((42, 0, 0, 5000), (2**40 + 17, 3, 9, 12345), (0, 31, 64, 777)):
        u, nz = eng_a.rng_fill(seed, view, draw, n)
        ou, onz = oracle.rng_fill(seed, view, draw, n)
        _eq(u, ou, "uniform")
        _eq(nz, onz, "normals")


def test_init_state_bit_exact(eng_a, scene_a):
    from oracle import oracle
    d, n, c = eng_a.init_state(42, 2, scene_a.depth_min, scene_a.depth_max)
    u, nz = oracle.rng_fill(42, 2, 0, scene_a.H * scene_a.W)
    H, W = scene_a.H, scene_a.W
    od, on, oc = oracle.init_state(u.reshape(H, W), nz[:, 0].reshape(H, W), nz[:, 1].reshape(H, W),
                                   scene_a.depth_min, scene_a.depth_max)
    _eq(d, od, "init depth")
    _eq(n, on, "init normal")
    _eq(c, oc, "init cost")


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
def test_box_stats_bit_exact(eng_a, scene_a, k):
    from oracle import oracle
    for v in (0, 3):
        m, var = eng_a.box_stats(v, k)
        om, ovar = oracle.box_stats(scene_a.grays[v], k)
        _eq(m, om, f"mean k{k}")
        _eq(var, ovar, f"var k{k}")


def _mixed_depth(scene, ref, seed):
    rng = np.random.default_rng(seed)
    d = np.exp(rng.uniform(np.log(scene.depth_min), np.log(scene.depth_max), (scene.H, scene.W))).astype(np.float32)
    d[:, scene.W // 2:] = scene.gt_depth[ref][:, scene.W // 2:]
    d[:5, :7] = np.float32(0.05)
    d[-6:, -9:] = np.float32(400.0)
    return d


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
@pytest.mark.parametrize("srcs", [[1, 3, 0, 4], [3, 1], [0, 1, 4], [1, 3, 0, 4, 2][:4]])
def test_eval_cost_bit_exact(eng_a, scene_a, k, srcs):
    ref = 2
    srcs = [s for s in srcs if s != ref]
    depth = _mixed_depth(scene_a, ref, 5)
    got = eng_a.eval_cost(ref, srcs, k, depth)
    want = scene_a.oracle_ctx(ref, srcs, k).patch_cost(depth)
    assert np.isposinf(want).any()
    _eq(got, want, f"cost k{k} S{len(srcs)}")


def test_eval_cost_vs_reference_golden(eng_a, scene_a):
    g = load_golden("g03_patch_cost")
    for k in (7, 11):
        for tag in ("s4", "s2"):
            srcs = list(g["srcs4"] if tag == "s4" else g["srcs2"])
            got = eng_a.eval_cost(int(g["ref"]), srcs, k, g["depth"])
            assert_cost_close(got, g[f"cost_k{k}_{tag}"], 1e-4, f"k{k} {tag}")


def test_confidence_bit_exact_and_golden(eng_a, scene_a):
    g = load_golden("g07_confidence")
    ref, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    got = eng_a.confidence(ref, srcs, k, g["depth"])
    _eq(got, scene_a.oracle_ctx(ref, srcs, k).confidence(g["depth"]), "confidence")
    assert np.mean(got != g["confidence"]) < 2e-3


@pytest.mark.parametrize("off", [(1, 0), (0, 1), (-1, 0), (0, -1)])
def test_propagate_step_bit_exact(eng_a, scene_a, off):
    g = load_golden("g04_propagate")
    ref, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    got = eng_a.propagate_step(ref, srcs, k, g["depth"], g["normal"], g["cost"], off[0], off[1], scene_a.depth_min)
    want = scene_a.oracle_ctx(ref, srcs, k).propagate_step(g["depth"], g["normal"], g["cost"], off[0], off[1],
                                                           scene_a.depth_min)
    for a, b, name in zip(got, want, ("depth", "normal", "cost")):
        _eq(a, b, f"propagate {off} {name}")
    assert (got[0] != g["depth"]).mean() > 0.01


def test_propagation_vs_reference_golden(eng_a, scene_a):
    g = load_golden("g04_propagate")
    ref, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    for tag, offs in (("even", [(1, 0), (0, 1)]), ("odd", [(-1, 0), (0, -1)])):
        d, n, c = g["depth"], g["normal"], g["cost"]
        for oy, ox in offs:
            d, n, c = eng_a.propagate_step(ref, srcs, k, d, n, c, oy, ox, scene_a.depth_min)
        assert np.mean(d == g[f"depth_{tag}"]) >= 0.995


@pytest.mark.parametrize("it", [0, 2])
def test_refine_step_bit_exact(eng_a, scene_a, it):
    from oracle import oracle
    g = load_golden("g05_refine")
    ref, srcs, k, seed = int(g["ref"]), list(g["srcs"]), int(g["patch"]), int(g["seed"])
    dr = np.float32((scene_a.depth_max - scene_a.depth_min) * 0.5 ** it)
    nr = np.float32(0.5 * 0.5 ** it)
    ctx = scene_a.oracle_ctx(ref, srcs, k)
    d, n, c = g["depth"], g["normal"], g["cost"]
    od, on, oc = d, n, c
    for s in range(2):
        draw = 1 + it * 2 + s
        d, n, c = eng_a.refine_step(ref, srcs, k, d, n, c, seed, ref, draw, dr, nr,
                                    scene_a.depth_min, scene_a.depth_max)
        u, nz = oracle.rng_fill(seed, ref, draw, scene_a.H * scene_a.W)
        od, on, oc = ctx.refine_step(od, on, oc, u, nz, dr, nr, scene_a.depth_min, scene_a.depth_max)
        _eq(d, od, f"refine it{it} s{s} depth")
        _eq(c, oc, f"refine it{it} s{s} cost")
        _eq(n, on, f"refine it{it} s{s} normal")
    assert np.all(c <= g["cost"])                                 # best cost never increases
    assert np.mean(d == g[f"depth_it{it}"]) >= 0.995              # reference golden


# ------------------------------------------------------------------ end to end ----
def test_patchmatch_bit_exact_vs_oracle_and_golden(eng_b, scene_b, amvs_mod):
    from amvs.engine import make_pm_params
    g = load_golden("g06_patchmatch_e2e")
    refs = [int(r) for r in g["refs"]]
    srcs = [list(g[f"srcs_{r}"]) for r in refs]
    p = make_pm_params(int(g["patch"]), int(g["iters"]), int(g["samples"]), scene_b.depth_min, scene_b.depth_max)
    depth, normal, conf = eng_b.patchmatch(refs, srcs, p, int(g["seed"]))      # both views in one batch
    t = eng_b.timing()
    assert t["sweep_launches"] == 3 * (2 + 4) and t["sweep_ms"] > 0
    assert t["pixel_hypotheses"] == 2 * scene_b.H * scene_b.W * 3 * 6
    for i, r in enumerate(refs):
        od, on, oc = scene_b.oracle_ctx(r, srcs[i], int(g["patch"])).patchmatch(
            int(g["iters"]), int(g["samples"]), scene_b.depth_min, scene_b.depth_max, int(g["seed"]), r)
        _eq(depth[i], od, f"view {r} depth")
        _eq(conf[i], oc, f"view {r} confidence")
        _eq(normal[i], on, f"view {r} normal")
        # reference: 1e-3 relative on identical RNG streams, as a pixel fraction
        rel = np.abs(depth[i] - g[f"depth_{r}"]) / g[f"depth_{r}"]
        assert np.mean(rel <= 1e-3) >= E2E_MIN_FRACTION
        hist_got = np.bincount(conf[i].astype(int).ravel(), minlength=5) / conf[i].size
        hist_ref = np.bincount(g[f"confidence_{r}"].astype(int).ravel(), minlength=5) / conf[i].size
        assert np.abs(hist_got - hist_ref).max() < CONF_HIST_TOL


def test_plane_sweep_bit_exact_and_golden(scene_c):
    g = load_golden("g11_plane_sweep")
    eng = scene_c.engine()
    try:
        ref, nbrs, k = int(g["ref"]), list(g["nbrs"]), int(g["patch"])
        depths = g["depths"].astype(np.float32)
        d, conf = eng.plane_sweep(ref, nbrs, depths, k, float(g["thresh"]))
        od, oc = scene_c.oracle_ctx(ref, nbrs, k).plane_sweep(depths, float(g["thresh"]))
        _eq(d, od, "sweep depth")
        _eq(conf, oc, "sweep confidence")
        assert np.mean(conf == g["confidence"]) > 0.995 and np.mean(d == g["depth_map"]) > 0.99
    finally:
        eng.close()


@pytest.mark.parametrize("shape", [(33, 59), (70, 117), (41, 200)])
def test_ragged_shapes_bit_exact(amvs_mod, shape):
    """Widths that are not multiples of the strip width, heights not multiples of the strip rows."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    H, W = shape
    sc = make_scene(4, H, W, seed=H)
    K = sc.camera.K.astype(np.float32)
    with amvs_mod.Engine(H, W, 4, K) as eng:
        for i in range(4):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        for k in (3, 5, 9, 11):
            p = make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max)
            depth, normal, conf = eng.patchmatch([1], [[0, 2, 3]], p, 9)
            ctx = oracle.ViewContext(K, sc.grays[1], sc.poses[1].R, sc.poses[1].t,
                                     [sc.grays[i] for i in (0, 2, 3)], [sc.poses[i].R for i in (0, 2, 3)],
                                     [sc.poses[i].t for i in (0, 2, 3)], k)
            od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 9, 1)
            _eq(depth[0], od, f"{shape} k{k} depth")
            _eq(conf[0], oc, f"{shape} k{k} conf")
            _eq(normal[0], on, f"{shape} k{k} normal")


@pytest.mark.parametrize("shape,nviews,k,S", [((9, 11), 3, 3, 2), ((5, 70), 4, 5, 3), ((66, 7), 4, 7, 3),
                                               ((40, 90), 7, 11, 6), ((30, 64), 7, 9, 5), ((24, 58), 7, 5, 6)])
def test_tiny_and_wide_source_sets_bit_exact(amvs_mod, shape, nviews, k, S):
    """Images smaller than one strip / one patch, single-column strips, and the widest source set
    with the largest patch (K=11, S=6: 64-bit validity history): still bit-exact."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    H, W = shape
    sc = make_scene(nviews, H, W, seed=H * 7 + W)
    g8 = [np.round(g * 255.0).astype(np.uint8) for g in sc.grays]
    grays = [g.astype(np.float32) / np.float32(255.0) for g in g8]          # 8-bit: packed sampling path
    K = sc.camera.K.astype(np.float32)
    ref = nviews // 2
    srcs = [i for i in range(nviews) if i != ref][:S]
    with amvs_mod.Engine(H, W, nviews, K) as eng:
        for i in range(nviews):
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        p = make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max)
        depth, normal, conf = eng.patchmatch([ref], [srcs], p, 3)
        ctx = oracle.ViewContext(K, grays[ref], sc.poses[ref].R, sc.poses[ref].t, [grays[i] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k)
        od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 3, ref)
        _eq(depth[0], od, "depth")
        _eq(conf[0], oc, "confidence")
        _eq(normal[0], on, "normal")
        if k in (5, 7) and S >= 2:
            depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, 9)).astype(np.float32)
            d, c = eng.plane_sweep(ref, srcs, depths, k, 0.8)
            wd, wc = ctx.plane_sweep(depths, 0.8)
            _eq(d, wd, "sweep depth")
            _eq(c, wc, "sweep confidence")


@pytest.mark.parametrize("shape,k,S,thr", [((37, 83), 5, 4, 0.8), ((70, 150), 5, 6, 0.35), ((41, 66), 7, 3, 0.8),
                                            ((33, 130), 3, 2, 0.05), ((52, 71), 9, 6, 0.93), ((29, 64), 11, 5, 0.8)])
def test_plane_sweep_float_images_bit_exact(amvs_mod, shape, k, S, thr):
    """The plane sweep on images that are NOT 8-bit exact (rendered floats: the float32 sampling path of the
    compiled plane-sweep kernels), planes from well in front of the scene to far behind it, several thresholds
    (the exact sweep's squared-comparison vote gate with its exact fall-back) -- and the same images once more
    through the packed path's switch (force_f32 on an 8-bit scene) must not change a bit.  Exact arithmetic: the
    fast mode samples the packed maps only and refuses other images (test_hip_fast_parity.py)."""
    from amvs.synthetic import make_scene
    from oracle import oracle
    mode = "exact"
    H, W = shape
    n = S + 1
    sc = make_scene(n, H, W, seed=H + W + k, arc_step_deg=14.0)
    K = sc.camera.K.astype(np.float32)
    ref = n // 2
    srcs = [i for i in range(n) if i != ref]
    planes = (1.0 / np.linspace(1 / (sc.depth_max * 3.0), 1 / (sc.depth_min * 0.3), 24)).astype(np.float32)
    with amvs_mod.Engine(H, W, n, K, mode=mode) as eng:
        for i in range(n):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "f32"
        ctx = oracle.ViewContext(K, sc.grays[ref], sc.poses[ref].R, sc.poses[ref].t, [sc.grays[i] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode=mode)
        for tile_rows in (0, 8):
            eng.set_sweep_tuning(tile_rows, 0)
            d, c = eng.plane_sweep(ref, srcs, planes, k, thr)
            wd, wc = ctx.plane_sweep(planes, thr)
            _eq(d, wd, f"{mode} k{k} S{S} rows {tile_rows} sweep depth")
            _eq(c, wc, f"{mode} k{k} S{S} rows {tile_rows} sweep confidence")
        assert c.max() > 0                                   # votes were cast: the comparison is not of empty maps
    g8 = [np.round(g * 255.0).astype(np.uint8).astype(np.float32) / np.float32(255.0) for g in sc.grays]
    with amvs_mod.Engine(H, W, n, K, mode=mode) as eng:
        for i in range(n):
            eng.set_view(i, g8[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        packed = eng.plane_sweep(ref, srcs, planes, k, thr)
        eng.set_sampling(force_f32=True)
        plain = eng.plane_sweep(ref, srcs, planes, k, thr)
        _eq(plain[0], packed[0], "forced float path: depth")
        _eq(plain[1], packed[1], "forced float path: confidence")
        ctx8 = oracle.ViewContext(K, g8[ref], sc.poses[ref].R, sc.poses[ref].t, [g8[i] for i in srcs],
                                  [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode=mode)
        wd, wc = ctx8.plane_sweep(planes, thr)
        _eq(packed[0], wd, "8-bit sweep depth")
        _eq(packed[1], wc, "8-bit sweep confidence")


@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("shape", [(2, 2), (2, 9), (9, 2), (3, 64), (64, 3), (2, 65), (5, 55), (4, 129)])
def test_smallest_images_every_patch_size(amvs_mod, mode, shape):
    """The smallest images the context accepts (2 x 2; the reference's grid normalisation divides by W - 1) up to
    one-strip-plus-one-column widths, with every compiled patch size and two run-time ones -- patches larger than
    the whole image, a single output row or column, strips whose halo is the entire strip: PatchMatch, plane sweep,
    cost and confidence maps, bit for bit."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    H, W = shape
    sc = make_scene(4, max(H, 16), max(W, 16), seed=H * 131 + W)          # rendered larger, cropped: poses stay sane
    grays = [(np.round(g[:H, :W] * 255.0).astype(np.uint8)).astype(np.float32) / np.float32(255.0) for g in sc.grays]
    grays = [np.ascontiguousarray(g) for g in grays]
    K = sc.camera.K.astype(np.float32)
    srcs = [0, 2, 3]
    planes = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, 7)).astype(np.float32)
    with amvs_mod.Engine(H, W, 4, K, mode=mode) as eng:
        for i in range(4):
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        for k in (3, 5, 7, 9, 11, 13, 31):
            ctx = oracle.ViewContext(K, grays[1], sc.poses[1].R, sc.poses[1].t, [grays[i] for i in srcs],
                                     [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode=mode)
            p = make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max)
            depth, normal, conf = eng.patchmatch([1], [srcs], p, 5)
            od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 5, 1)
            tag = f"{mode} {H}x{W} k{k}"
            _eq(depth[0], od, tag + " depth")
            _eq(conf[0], oc, tag + " confidence")
            _eq(normal[0], on, tag + " normal")
            d, c = eng.plane_sweep(1, srcs, planes, k, 0.5)
            wd, wc = ctx.plane_sweep(planes, 0.5)
            _eq(d, wd, tag + " sweep depth")
            _eq(c, wc, tag + " sweep confidence")
            flat = np.full((H, W), np.float32(0.5 * (sc.depth_min + sc.depth_max)), np.float32)
            _eq(eng.eval_cost(1, srcs, k, flat), ctx.patch_cost(flat), tag + " cost")
            _eq(eng.confidence(1, srcs, k, flat), ctx.confidence(flat), tag + " confidence of a flat map")


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_wide_baselines_bit_exact(amvs_mod, seed):
    """Random small shapes, patch sizes and source counts with wide baselines and a depth range that
    throws many projections far outside the sources: the packed maps' zero border and origin clamp
    (no tap masks) against the oracle's masked 4-tap sampler, PatchMatch and plane sweep."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    rng = np.random.default_rng(1000 + seed)
    H, W = int(rng.integers(8, 80)), int(rng.integers(8, 150))
    k = int(rng.choice([3, 5, 7]))
    S = int(rng.integers(2, 5))
    nviews = S + 1
    sc = make_scene(nviews, H, W, seed=seed + 11, arc_step_deg=float(rng.uniform(12.0, 30.0)))
    grays = [(np.round(g * 255.0).astype(np.uint8)).astype(np.float32) / np.float32(255.0) for g in sc.grays]
    K = sc.camera.K.astype(np.float32)
    ref = int(rng.integers(0, nviews))
    srcs = [i for i in range(nviews) if i != ref]
    dmin, dmax = np.float32(sc.depth_min * 0.2), np.float32(sc.depth_max * 4.0)
    with amvs_mod.Engine(H, W, nviews, K) as eng:
        for i in range(nviews):
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs"
        depth, normal, conf = eng.patchmatch([ref], [srcs], make_pm_params(k, 2, 3, dmin, dmax), 40 + seed)
        ctx = oracle.ViewContext(K, grays[ref], sc.poses[ref].R, sc.poses[ref].t, [grays[i] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k)
        od, on, oc = ctx.patchmatch(2, 3, dmin, dmax, 40 + seed, ref)
        tag = f"seed {seed}: {H}x{W} k{k} S{S}"
        _eq(depth[0], od, tag + " depth")
        _eq(conf[0], oc, tag + " confidence")
        _eq(normal[0], on, tag + " normal")
        if k in (5, 7):
            planes = (1.0 / np.linspace(1 / dmax, 1 / dmin, 12)).astype(np.float32)
            d, c = eng.plane_sweep(ref, srcs, planes, k, 0.8)
            wd, wc = ctx.plane_sweep(planes, 0.8)
            _eq(d, wd, tag + " sweep depth")
            _eq(c, wc, tag + " sweep confidence")


# --------------------------------------------------- full-size, size-independent ----
def test_full_size_invariants(amvs_mod):
    """1920x1080 (BASELINE config 3 resolution): results do not depend on the strip height,
    on batching, or on the run; best cost is monotone; the ground-truth depth scores far
    better than a random one."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    H, W, n = 1080, 1920, 5
    sc = make_scene(n, H, W, seed=21)
    K = sc.camera.K.astype(np.float32)
    with amvs_mod.Engine(H, W, n, K) as eng:
        for i in range(n):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        srcs = {2: [1, 3, 0, 4], 1: [0, 2, 3, 4]}
        base = None
        for tile_rows in (64, 8, 24):
            p = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max, tile_rows=tile_rows)
            out = eng.patchmatch([2], [srcs[2]], p, 5)
            if base is None:
                base = out
            else:
                for a, b, name in zip(out, base, ("depth", "normal", "confidence")):
                    _eq(a, b, f"tile_rows {tile_rows} {name}")
        p = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max)
        both = eng.patchmatch([1, 2], [srcs[1], srcs[2]], p, 5)
        assert eng.last_views_per_launch() == 2
        split = eng.patchmatch([1, 2], [srcs[1], srcs[2]],
                               make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max, views_per_launch=1), 5)
        assert eng.last_views_per_launch() == 1
        for a, b, name in zip(split, both, ("depth", "normal", "confidence")):
            _eq(a, b, f"views_per_launch=1 {name}")
        for a, b, name in zip(both, base, ("depth", "normal", "confidence")):
            _eq(a[1], b[0], f"batched {name}")
        assert np.isfinite(base[0]).all() and base[0].min() >= np.float32(sc.depth_min) \
            and base[0].max() <= np.float32(sc.depth_max)
        assert eng.sampling_mode() == "f32"            # rendered floats are not 8-bit exact
        # unit normals, or the exact zero normal that propagation pulls in from beyond the
        # image border (F.pad of the normal map, mvs_patchmatch.py:432-442)
        nrm = np.linalg.norm(base[1], axis=-1)
        assert np.all((np.abs(nrm - 1) < 1e-5) | (nrm == 0))
        assert np.mean(nrm == 0) < 0.01
        c_gt = eng.eval_cost(2, srcs[2], 7, sc.depths[2])
        c_rand = eng.eval_cost(2, srcs[2], 7, np.full((H, W), sc.depth_max * 0.9, np.float32))
        inner = (slice(100, -100), slice(200, -200))
        assert np.nanmean(c_gt[inner][np.isfinite(c_gt[inner])]) < 0.1
        assert np.nanmean(c_rand[inner][np.isfinite(c_rand[inner])]) > 0.5
        conf_gt = eng.confidence(2, srcs[2], 7, sc.depths[2])
        assert np.mean(conf_gt[inner] >= 3) > 0.9


def test_packed_and_float_sampling_agree(scene_b):
    """8-bit scenes take the packed row-pair path; forcing the float32 path (amvs_set_sampling) must
    not change a single bit in exact mode (and the golden scenes do exercise the packed path)."""
    from amvs.engine import make_pm_params
    p = make_pm_params(7, 2, 3, scene_b.depth_min, scene_b.depth_max)
    eng = scene_b.engine()
    assert eng.sampling_mode() == "u8-pairs"
    packed = eng.patchmatch([1, 3], [[0, 2, 3, 4], [1, 2, 4, 0]], p, 11)
    eng.set_sampling(force_f32=True)
    assert eng.sampling_mode() == "f32"
    plain = eng.patchmatch([1, 3], [[0, 2, 3, 4], [1, 2, 4, 0]], p, 11)
    eng.close()
    for a, b, name in zip(packed, plain, ("depth", "normal", "confidence")):
        _eq(a, b, name)


def test_non_8bit_images_use_float_path(amvs_mod):
    from amvs.synthetic import make_scene
    sc = make_scene(3, 40, 70, seed=2)
    with amvs_mod.Engine(40, 70, 3, sc.camera.K.astype(np.float32)) as eng:
        for i in range(3):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "f32"


def test_4k_sweep_runs_and_is_tiling_invariant(amvs_mod):
    """BASELINE config 5 resolution (3840x2160): indices, strides and the strip grid hold up."""
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    H, W = 2160, 3840
    sc = make_scene(3, H, W, seed=4)
    with amvs_mod.Engine(H, W, 3, sc.camera.K.astype(np.float32)) as eng:
        for i in range(3):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        outs = []
        for tr in (32, 16):
            p = make_pm_params(7, 1, 1, sc.depth_min, sc.depth_max, tile_rows=tr)
            outs.append(eng.patchmatch([1], [[0, 2]], p, 1))
        for a, b, name in zip(outs[0], outs[1], ("depth", "normal", "confidence")):
            _eq(a, b, f"4K {name}")
        d = outs[0][0][0]
        assert np.isfinite(d).all() and d.min() >= np.float32(sc.depth_min) and d.max() <= np.float32(sc.depth_max)
        # the last rows / columns were written (not left at the allocation's garbage)
        assert outs[0][2][0, -1, -1] in (0.0, 1.0, 2.0) and outs[0][2][0, -1, 0] in (0.0, 1.0, 2.0)


# ------------------------------------------------------------------ fusion / filter ---
def test_device_fusion_matches_reference_golden(eng_b, scene_b, amvs_mod):
    """amvs_fuse_filter against the clouds the REFERENCE produced (g10): bit-identical points and
    colours, raw and filtered."""
    g = load_golden("g10_fuse_filter")
    refs = [int(r) for r in g["refs"]]
    depth = np.stack([scene_b.gt_depth[r] for r in refs])
    cols = np.stack([scene_b.colors[r] for r in refs])
    K_inv = np.linalg.inv(scene_b.K)
    poses = [(scene_b.R[r], scene_b.t[r]) for r in refs]
    pts, rgb, raw = eng_b.fuse_filter(depth, g["confidence"], cols, K_inv, poses, 3, do_filter=False)
    assert raw == len(g["points"])
    assert np.array_equal(pts, g["points"]) and np.array_equal(rgb, g["colors"])
    fp, fc, raw2 = eng_b.fuse_filter(depth, g["confidence"], cols, K_inv, poses, 3, do_filter=True)
    assert raw2 == raw
    assert np.array_equal(fp, g["f_points"]) and np.array_equal(fc, g["f_colors"])


def test_device_fusion_matches_numpy_on_sweep_output(amvs_mod):
    """Fusion of real sweep output at a larger size (odd and even point counts, duplicate voxels):
    device cloud == NumPy restatement of mvs_patchmatch.py:536-588."""
    from amvs.core.mvs_patchmatch import DepthNormalMap, PatchMatchMVS
    from amvs.synthetic import make_scene
    sc = make_scene(5, 270, 480, seed=9)
    pm = PatchMatchMVS(amvs_mod.Camera(K=sc.camera.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7)
    with amvs_mod.Engine(270, 480, 5, sc.camera.K.astype(np.float32)) as eng:
        for i in range(5):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        for min_views, drop in ((3, 0), (2, 1), (4, 0)):
            maps, proc = {}, {}
            for r in (1, 2, 3):
                conf = eng.confidence(r, [i for i in range(5) if i != r], 7, sc.depths[r])
                if drop:
                    conf.flat[7] = 0.0                      # flip the parity of the point count
                maps[r] = DepthNormalMap(depth=sc.depths[r], normal=None, confidence=conf)
                proc[r] = {"color": sc.colors[r]}
            pm.min_views = min_views
            want_p, want_c = pm._fuse_depth_maps(maps, proc, sc.poses)
            assert len(want_p) > 1000
            want_fp, want_fc = pm._filter_points(want_p, want_c)
            ids = list(maps)
            got_p, got_c, raw = eng.fuse_filter(np.stack([maps[i].depth for i in ids]),
                                                np.stack([maps[i].confidence for i in ids]),
                                                np.stack([sc.colors[i] for i in ids]), np.linalg.inv(pm.K_scaled),
                                                [(sc.poses[i].R, sc.poses[i].t) for i in ids], min_views, True)
            assert raw == len(want_p)
            assert len(want_fp) < len(want_p)                # the filter and the voxel grid removed points
            assert np.array_equal(got_p, want_fp) and np.array_equal(got_c, want_fc)
    empty = DepthNormalMap(depth=sc.depths[0], normal=None, confidence=np.zeros((270, 480), np.float32))
    pm._engine = None
    p0, c0 = pm._fuse_depth_maps({0: empty}, {0: {"color": sc.colors[0]}}, sc.poses)
    assert p0.shape == (0, 3)


# ------------------------------------------------------------------ error paths ---
def _knn_clouds():
    rng = np.random.default_rng(77)
    clouds = {}
    # a reconstructed-surface-like cloud: two noisy sheets
    xy = rng.uniform(-2.0, 2.0, (20000, 2))
    z = 0.15 * np.sin(1.3 * xy[:, 0]) * np.cos(1.7 * xy[:, 1]) + rng.normal(0, 0.004, 20000)
    sheet = np.column_stack([xy, z])
    clouds["sheets"] = np.vstack([sheet, sheet[:5000] * [1.0, 1.0, -1.0] + [0.0, 0.0, 1.0]])
    # exact duplicates (several zero distances) and far outliers (long shell walks)
    dup = sheet[:3000].copy()
    dup[100:160] = dup[7]
    dup[-5:] = [[40.0, 0, 0], [0, -55.0, 3], [9.0, 9.0, 9.0], [-30.0, 30.0, 0.5], [40.001, 0, 0]]
    clouds["duplicates+outliers"] = dup
    clouds["few points"] = rng.normal(size=(45, 3))          # smallest size scikit-learn still uses its KD-tree for
    line = np.zeros((400, 3))
    line[:, 0] = np.sort(rng.uniform(0, 1, 400))           # degenerate extents in y and z
    clouds["line"] = line
    clouds["volume float32 values"] = rng.uniform(-1, 1, (6000, 3)).astype(np.float32).astype(np.float64)
    # strongly non-uniform: a tight cluster that sets the cell size and a wide halo a third of the
    # points live in -- more grid levels than three are needed before few enough queries are left
    # for the per-query scans
    clouds["cluster+halo"] = np.vstack([rng.normal(0, 0.01, (20000, 3)), rng.uniform(-50.0, 50.0, (10000, 3))])
    return clouds


def test_knn_mean_distance_matches_sklearn(eng_a):
    """amvs_knn_mean_distance against the reference's own expression (dense_stereo.py:456-460:
    scikit-learn kneighbors + np.mean over distances[:, 1:]): bit-identical float64."""
    NearestNeighbors = pytest.importorskip("sklearn.neighbors").NearestNeighbors
    for name, pts in _knn_clouds().items():
        for k in {"sheets": (20, 8), "duplicates+outliers": (20, 10, 16, 32), "cluster+halo": (20, 32)}.get(name, (20,)):
            dists, _ = NearestNeighbors(n_neighbors=k).fit(pts).kneighbors(pts)
            want = np.mean(dists[:, 1:], axis=1)
            got = eng_a.knn_mean_distance(pts, k)
            assert got.dtype == np.float64 and got.shape == want.shape
            bad = got != want
            assert not bad.any(), f"{name}, k={k}: {int(bad.sum())} of {len(want)} means differ, " \
                                  f"max |diff| {np.abs(got - want).max():.3e}"
    # below n = 2k + 2 scikit-learn switches to its brute-force kernel (|x|^2 + |y|^2 - 2xy, a few
    # ulps off the direct expression): same neighbours, values equal to rounding only
    tiny = np.random.default_rng(3).normal(size=(21, 3))
    dists, _ = NearestNeighbors(n_neighbors=20).fit(tiny).kneighbors(tiny)
    np.testing.assert_allclose(eng_a.knn_mean_distance(tiny, 20), np.mean(dists[:, 1:], axis=1), rtol=1e-12)


def test_stereo_outlier_filter_device_equals_host(eng_a, amvs_mod):
    """DenseStereoReconstructor._filter_outliers with the device neighbour search selects exactly
    the points the scikit-learn path selects."""
    pytest.importorskip("sklearn.neighbors")
    from amvs.core.dense_stereo import DenseStereoReconstructor
    pts = _knn_clouds()["sheets"]
    cols = (np.arange(len(pts) * 3) % 251).astype(np.uint8).reshape(-1, 3)
    cam = amvs_mod.Camera(K=np.array([[500.0, 0, 320], [0, 500.0, 240], [0, 0, 1]]), dist=np.zeros(5))
    rec = DenseStereoReconstructor(cam, scale=1.0)
    host_p, host_c = rec._filter_outliers(pts, cols)             # no engine yet: scikit-learn
    rec._engine = eng_a
    dev_p, dev_c = rec._filter_outliers(pts, cols)
    rec._engine = None
    assert 0 < len(host_p) < len(pts)
    assert np.array_equal(dev_p, host_p) and np.array_equal(dev_c, host_c)


def test_stereo_reconstruct_device_filter_equals_sklearn_filter(amvs_mod):
    """DenseStereoReconstructor.reconstruct end to end on the GPU: the cloud with the device
    neighbour search equals the cloud with the reference's scikit-learn search, point for point."""
    pytest.importorskip("sklearn.neighbors")
    from amvs.core.dense_stereo import DenseStereoReconstructor
    from amvs.synthetic import make_scene
    sc = make_scene(5, 120, 160, seed=21)
    images = [{"image": np.ascontiguousarray(c[:, :, ::-1])} for c in sc.colors]
    poses = dict(sc.poses) if isinstance(sc.poses, dict) else {i: p for i, p in enumerate(sc.poses)}
    clouds = []
    for device_filter in (True, False):
        rec = DenseStereoReconstructor(sc.camera, scale=1.0, device_filter=device_filter)
        clouds.append(rec.reconstruct(images, poses, max_pairs=30))
    (p_dev, c_dev), (p_host, c_host) = clouds
    assert len(p_host) > 100
    assert np.array_equal(p_dev, p_host) and np.array_equal(c_dev, c_host)


def test_stereo_big_cloud_random_subsample_on_the_device(amvs_mod, monkeypatch, capsys):
    """Round 4: clouds above 500 000 points, which the reference sub-samples with an unseeded np.random.choice before
    its outlier filter (dense_stereo.py:449-451).  The class makes the same draw (in kept buffers: the indices
    np.random.choice returns for the same generator state, tests/test_host_logic.py) and takes the sample on the device
    (amvs_cloud_take), so the 2 M-point cloud never travels to the host; for the SAME generator state it must return
    exactly the cloud of the host path, which calls np.random.choice itself as the reference does (fetch,
    points[chosen], neighbour statistic, numpy selection, numpy voxel grid)."""
    import torch
    from amvs.synthetic import make_scene
    sc = make_scene(12, 756, 1008, seed=8, device="cuda" if torch.cuda.is_available() else "cpu")
    images = sc.images()
    cam = amvs_mod.Camera(K=sc.camera.K.copy(), dist=np.zeros(5))
    draws = []
    real_choice, real_shuffle = np.random.choice, np.random.shuffle

    def counting_choice(*a, **k):
        draws.append(("choice", a[0]))
        return real_choice(*a, **k)

    def counting_shuffle(x):
        draws.append(("shuffle", len(x)))
        return real_shuffle(x)
    monkeypatch.setattr(np.random, "choice", counting_choice)
    monkeypatch.setattr(np.random, "shuffle", counting_shuffle)
    out = {}
    for where in ("device", "host"):
        ds = amvs_mod.DenseStereoReconstructor(cam, scale=1.0, num_depths=32, min_views=2)
        ds._subsample_on_host = where == "host"
        np.random.seed(1234)
        out[where] = ds.reconstruct(images, sc.poses)
        ds._engine.close()
    capsys.readouterr()
    assert len(draws) == 2 and draws[0][0] == "shuffle" and draws[1][0] == "choice" and draws[0][1] == draws[1][1] > 500000, draws
    assert len(out["device"][0]) > 10000
    assert np.array_equal(out["device"][0], out["host"][0]) and np.array_equal(out["device"][1], out["host"][1])
    # the entry point refuses indices outside the resident cloud
    from amvs._lib import AmvsError
    sc2 = GoldenScene("scene_c")
    with sc2.engine() as eng:
        with pytest.raises(AmvsError):
            eng.cloud_take([0, 1, 2])                      # no resident cloud


@pytest.mark.parametrize("scale", [1.0, 0.5, 0.25, 0.3, 0.6])
def test_device_image_preparation_equals_host_restatement(amvs_mod, scale):
    """amvs_set_view_bgr8 (upload the 8-bit BGR image, cv.resize + cvtColor arithmetic on the GPU) against
    core/imageprep.py: identical resized colour image, identical gray map (checked through the box
    statistics and a cost evaluation of views prepared either way)."""
    from amvs.core.imageprep import prepare_view
    from amvs.synthetic import make_scene
    sc = make_scene(4, 90, 130, seed=31)
    rng = np.random.default_rng(5)
    imgs = [np.clip(c.astype(np.int16) + rng.integers(-20, 21, c.shape), 0, 255).astype(np.uint8) for c in sc.colors]
    prep = [prepare_view(im, scale) for im in imgs]
    H, W = prep[0]["shape"]
    K = sc.camera.K.copy()
    K[:2] *= scale
    K = K.astype(np.float32)
    with amvs_mod.Engine(H, W, 4, K) as dev, amvs_mod.Engine(H, W, 4, K) as host:
        for i in range(4):
            color = dev.set_view_bgr8(i, imgs[i], sc.poses[i].R, sc.poses[i].t)
            assert np.array_equal(color, prep[i]["color"]), f"view {i}: resized colour image differs"
            host.set_view(i, prep[i]["gray"], sc.poses[i].R, sc.poses[i].t)
        assert dev.sampling_mode() == host.sampling_mode() == "u8-pairs"
        for v in (0, 3):
            md, vd = dev.box_stats(v, 5)
            mh, vh = host.box_stats(v, 5)
            _eq(md, mh, f"scale {scale} view {v} mean")
            _eq(vd, vh, f"scale {scale} view {v} variance")
        d = np.full((H, W), 5.0, np.float32)
        _eq(dev.eval_cost(1, [0, 2, 3], 5, d), host.eval_cost(1, [0, 2, 3], 5, d), f"scale {scale} cost")


def test_reconstruct_device_prep_equals_host_prep(amvs_mod):
    """PatchMatchMVS.reconstruct and DenseStereoReconstructor.reconstruct at the CLI's scale 0.25 with the
    images prepared on the GPU and on the host: the same cloud."""
    from amvs.core.dense_stereo import DenseStereoReconstructor
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    from amvs.synthetic import make_scene
    sc = make_scene(5, 288, 384, seed=8)
    images = [{"image": c} for c in sc.colors]
    clouds = []
    for device_prep in (True, False):
        pm = PatchMatchMVS(sc.camera, scale=0.25, patch_size=5, num_iterations=2, num_samples=2, min_views=2, seed=3,
                           device_prep=device_prep)
        pm._estimate_depth_range = lambda poses, sparse: None
        pm.depth_min, pm.depth_max = sc.depth_min, sc.depth_max
        clouds.append(pm.reconstruct(images, dict(sc.poses)))
    assert len(clouds[0][0]) > 0
    assert np.array_equal(clouds[0][0], clouds[1][0]) and np.array_equal(clouds[0][1], clouds[1][1])
    st = []
    for device_prep in (True, False):
        rec = DenseStereoReconstructor(sc.camera, scale=0.25, num_depths=16, min_views=2, device_prep=device_prep)
        st.append(rec.reconstruct(images, dict(sc.poses), max_pairs=30))
    assert len(st[0][0]) > 0
    assert np.array_equal(st[0][0], st[1][0]) and np.array_equal(st[0][1], st[1][1])


def test_stereo_post_steps_on_device_match_reference_golden(scene_c):
    """Device back-projection, neighbour statistic and voxel down-sampling of the stereo path against
    the clouds the REFERENCE produced (g12 from g11's maps and from the ground-truth depth):
    bit-identical points and colours."""
    pytest.importorskip("sklearn.neighbors")
    g11, g12 = load_golden("g11_plane_sweep"), load_golden("g12_stereo_post")
    ref = int(g11["ref"])
    K_inv = np.linalg.inv(scene_c.K)
    pose = [(scene_c.R[ref], scene_c.t[ref])]
    cols = scene_c.colors[ref][None]
    eng = scene_c.engine()
    try:
        # _backproject of the reference's own plane-sweep maps
        counts, total, pts, rgb = eng.stereo_backproject(cols, K_inv, pose, 2.5, depth=g11["depth_map"][None],
                                                         conf=g11["confidence"][None], fetch=True)
        assert counts == [total] and total == len(g12["bp_points"])
        assert np.array_equal(pts, g12["bp_points"]) and np.array_equal(rgb, g12["bp_colors"])
        # ground-truth depth, every pixel confident: back-projection, voxel grid, outlier statistic
        conf4 = np.full((1, scene_c.H, scene_c.W), 4.0, np.float32)
        counts, total, pts, rgb = eng.stereo_backproject(cols, K_inv, pose, 2.5, depth=scene_c.gt_depth[ref][None],
                                                         conf=conf4, fetch=True)
        assert np.array_equal(pts, g12["gt_points"]) and np.array_equal(rgb, g12["gt_colors"])
        mean_d = eng.cloud_knn_mean_distance(total, 20)
        keep = mean_d < np.mean(mean_d) + 2.0 * np.std(mean_d)
        assert np.array_equal(pts[keep], g12["out_points"]) and np.array_equal(rgb[keep], g12["out_colors"])
        m = eng.cloud_voxel_downsample(0.02)
        vp, vc = eng.fetch_cloud(m)
        assert np.array_equal(vp, g12["vox_points"]) and np.array_equal(vc, g12["vox_colors"])
        # mask + voxel grid in one call == numpy on the masked cloud
        from amvs.core.dense_stereo import DenseStereoReconstructor
        eng.stereo_backproject(cols, K_inv, pose, 2.5, depth=scene_c.gt_depth[ref][None], conf=conf4)
        m = eng.cloud_voxel_downsample(0.02, keep)
        got_p, got_c = eng.fetch_cloud(m)
        rec = DenseStereoReconstructor.__new__(DenseStereoReconstructor)
        want_p, want_c = rec._voxel_down_sample(pts[keep], rgb[keep], 0.02)
        assert np.array_equal(got_p, want_p) and np.array_equal(got_c, want_c)
    finally:
        eng.close()


def test_plane_sweep_batch_keeps_maps_resident(scene_d):
    """amvs_plane_sweep_batch == per-view amvs_plane_sweep, and the resident maps back-project to the
    same cloud as the host copies (per-view counts included)."""
    g = load_golden("g16_plane_sweep_s6")
    depths = g["depths"].astype(np.float32)
    refs = [3, 1, 5]
    nbrs = [[i for i in range(scene_d.n) if i != r][:6] for r in refs]
    K_inv = np.linalg.inv(scene_d.K)
    eng = scene_d.engine("fast")
    try:
        singles = [eng.plane_sweep(r, nb, depths, 5, 0.8) for r, nb in zip(refs, nbrs)]
        eng.plane_sweep_batch(refs, nbrs, depths, 5, 0.8)
        d, c = eng.fetch_sweep_maps(0, 3)
        for i in range(3):
            _eq(d[i], singles[i][0], f"batch view {i} depth")
            _eq(c[i], singles[i][1], f"batch view {i} confidence")
        cols = np.stack([scene_d.colors[r] for r in refs])
        poses = [(scene_d.R[r], scene_d.t[r]) for r in refs]
        counts_r, total_r, p_r, c_r = eng.stereo_backproject(cols, K_inv, poses, 1.5, fetch=True)      # resident maps
        counts_h, total_h, p_h, c_h = eng.stereo_backproject(cols, K_inv, poses, 1.5, depth=d, conf=c, fetch=True)
        assert counts_r == counts_h and total_r == total_h == sum(counts_r) and total_r > 100
        assert counts_r == [int(((c[i] >= 1.5) & (d[i] > 0)).sum()) for i in range(3)]
        assert np.array_equal(p_r, p_h) and np.array_equal(c_r, c_h)
    finally:
        eng.close()


def test_error_paths(eng_a, scene_a, amvs_mod):
    from amvs._lib import AmvsError
    d = scene_a.gt_depth[2]
    with pytest.raises(AmvsError, match="patch_size"):
        eng_a.eval_cost(2, [1, 3], 33, d)              # (odd sizes up to 31 run: tests/test_hip_generic_patch.py)
    with pytest.raises(AmvsError, match="patch_size"):
        eng_a.eval_cost(2, [1, 3], 6, d)
    with pytest.raises(AmvsError, match="n_src"):
        eng_a.eval_cost(2, [1], 7, d)
    with pytest.raises(AmvsError, match="not uploaded"):
        eng_a.eval_cost(2, [1, 17], 7, d)
    with pytest.raises(AmvsError):
        amvs_mod.Engine(0, 10, 3, np.eye(3))


def test_resident_colour_entry_points_need_device_prepared_views(amvs_mod):
    """amvs_fuse_filter_views / amvs_stereo_backproject_views read the colour images amvs_set_view_bgr8
    leaves on the device; a view uploaded as a gray map has none, and the call says so.  With device-
    prepared views the cloud equals the one fused from host colour arrays."""
    import torch
    from amvs.synthetic import make_scene
    sc = make_scene(3, 40, 56, seed=5)
    K = sc.camera.K.astype(np.float32)
    K_inv = np.linalg.inv(sc.camera.K)
    dev = torch.device("cuda", 0)
    depth = torch.from_numpy(np.stack([sc.depths[i] for i in range(3)]).astype(np.float32)).to(dev)
    conf = torch.full((3, 40, 56), 3.0, dtype=torch.float32, device=dev)
    view_poses = [(sc.poses[i].R, sc.poses[i].t) for i in range(3)]
    with amvs_mod.Engine(40, 56, 3, K) as eng:
        for i in range(3):
            eng.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        with pytest.raises(amvs_mod.AmvsError, match="no resident colour image"):
            eng.fuse_filter_views([0, 1, 2], depth.data_ptr(), conf.data_ptr(), K_inv, view_poses, 3)
        bgr = [np.ascontiguousarray(sc.colors[i][:, :, ::-1]) for i in range(3)]
        for i in range(3):
            eng.set_view_bgr8(i, bgr[i], sc.poses[i].R, sc.poses[i].t, want_color=False)
        torch.cuda.synchronize()
        p1, c1, raw1 = eng.fuse_filter_views([2, 0, 1], depth[[2, 0, 1]].contiguous().data_ptr(),
                                             conf.data_ptr(), K_inv, [view_poses[j] for j in (2, 0, 1)], 3)
        p2, c2, raw2 = eng.fuse_filter(depth[[2, 0, 1]].cpu().numpy(), conf.cpu().numpy(), np.stack([bgr[j] for j in (2, 0, 1)]),
                                       K_inv, [view_poses[j] for j in (2, 0, 1)], 3)
    assert raw1 == raw2 and np.array_equal(p1, p2) and np.array_equal(c1, c2)
