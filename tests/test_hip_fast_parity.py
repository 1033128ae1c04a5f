"""HIP kernels in the FAST arithmetic (AMVS_MODE_FAST) through the C ABI.

Bar: BIT-EXACT against the CPU oracle's fast mode (oracle/amvs_oracle.c orc_ctx_set_mode(1)), which
tests/test_oracle_modes_golden.py pins against the reference's golden vectors; the same golden
tolerances are re-checked here on the HIP outputs (>= 98 % of depths within 1e-3 relative,
confidence histogram within 1 %, Chamfer distance of the fused cloud).  This is the mode bench.py
times, so the full-size tests of tests/test_hip_fullsize_parity.py run it as well.
"""
import numpy as np
import pytest

from conftest import CHAMFER_TOL, CONF_HIST_TOL, E2E_MIN_FRACTION, chamfer, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amvs_mod():
    import amvs
    return amvs


@pytest.fixture(scope="module")
def feng_a(scene_a, amvs_mod):
    eng = scene_a.engine("fast")
    yield eng
    eng.close()


@pytest.fixture(scope="module")
def feng_b(scene_b, amvs_mod):
    eng = scene_b.engine("fast")
    yield eng
    eng.close()


def _eq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ " \
                       f"(first at {np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r})"


def _mixed_depth(scene, ref, seed):
    rng = np.random.default_rng(seed)
    d = np.exp(rng.uniform(np.log(scene.depth_min), np.log(scene.depth_max), (scene.H, scene.W))).astype(np.float32)
    d[:, scene.W // 2:] = scene.gt_depth[ref][:, scene.W // 2:]
    d[:5, :7] = np.float32(0.05)
    d[-6:, -9:] = np.float32(400.0)
    return d


# ------------------------------------------------------------------ stages --------
@pytest.mark.parametrize("mode", ["exact", "fast"])
@pytest.mark.parametrize("bounds", [0, 1, 2])
def test_sampling_stage_bit_exact(scene_a, mode, bounds):
    """The stage before the box filter (projection, validity, bilinear sample), both modes, the
    three validity rules; includes depths that throw projections far outside the sources, z <= 0.1
    and non-finite depths."""
    ref, srcs = 2, [1, 3, 0, 4]
    depth = _mixed_depth(scene_a, ref, 5)
    depth[10, 10:14] = [np.inf, np.nan, 0.0, -3.0]
    depth[11, 10:12] = [1e-30, 1e30]
    eng = scene_a.engine(mode)
    try:
        got, gvalid = eng.sample_sources(ref, srcs, 7, depth, bounds)
    finally:
        eng.close()
    ctx = scene_a.oracle_ctx(ref, srcs, 7, mode)
    for s in range(len(srcs)):
        want, wvalid = ctx.sample(s, depth, bounds)
        _eq(got[s], want, f"{mode} bounds {bounds} source {s} sample")
        assert np.array_equal(gvalid[s], wvalid), f"{mode} bounds {bounds} source {s} validity"


@pytest.mark.parametrize("k", [3, 5, 7, 9, 11])
@pytest.mark.parametrize("srcs", [[1, 3, 0, 4], [3, 1], [0, 1, 4]])
def test_eval_cost_bit_exact(feng_a, scene_a, k, srcs):
    ref = 2
    depth = _mixed_depth(scene_a, ref, 5)
    got = feng_a.eval_cost(ref, srcs, k, depth)
    want = scene_a.oracle_ctx(ref, srcs, k, "fast").patch_cost(depth)
    assert np.isposinf(want).any()
    _eq(got, want, f"fast cost k{k} S{len(srcs)}")


def test_eval_cost_vs_reference_golden(feng_a):
    """Against the REFERENCE (g03, g15): finite costs within 1e-4 except knife-edge validity pixels."""
    g03, g15 = load_golden("g03_patch_cost"), load_golden("g15_patch_cost_k359")
    ref, srcs = int(g15["ref"]), list(g15["srcs"])
    for k in (3, 5, 7, 9, 11):
        want = g15[f"cost_k{k}"] if k in (3, 5, 9) else g03[f"cost_k{k}_s4"]
        got = feng_a.eval_cost(ref, srcs, k, g15["depth"])
        flips = int((np.isposinf(got) != np.isposinf(want)).sum() + (np.isnan(got) != np.isnan(want)).sum())
        fin = np.isfinite(got) & np.isfinite(want)
        err = np.abs(got[fin] - want[fin])
        assert flips + int((err > 1e-4).sum()) <= 3, f"k{k}"
        assert np.median(err) < 5e-6


def test_confidence_bit_exact_and_golden(feng_a, scene_a):
    g = load_golden("g07_confidence")
    ref, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    got = feng_a.confidence(ref, srcs, k, g["depth"])
    _eq(got, scene_a.oracle_ctx(ref, srcs, k, "fast").confidence(g["depth"]), "fast confidence")
    assert np.mean(got != g["confidence"]) < 1e-3


@pytest.mark.parametrize("off", [(1, 0), (0, 1), (-1, 0), (0, -1)])
def test_propagate_step_bit_exact(feng_a, scene_a, off):
    g = load_golden("g04_propagate")
    ref, srcs, k = int(g["ref"]), list(g["srcs"]), int(g["patch"])
    got = feng_a.propagate_step(ref, srcs, k, g["depth"], g["normal"], g["cost"], off[0], off[1], scene_a.depth_min)
    want = scene_a.oracle_ctx(ref, srcs, k, "fast").propagate_step(g["depth"], g["normal"], g["cost"], off[0], off[1],
                                                                   scene_a.depth_min)
    for a, b, name in zip(got, want, ("depth", "normal", "cost")):
        _eq(a, b, f"fast propagate {off} {name}")


@pytest.mark.parametrize("it", [0, 2])
def test_refine_step_bit_exact(feng_a, scene_a, it):
    from oracle import oracle
    g = load_golden("g05_refine")
    ref, srcs, k, seed = int(g["ref"]), list(g["srcs"]), int(g["patch"]), int(g["seed"])
    dr = np.float32((scene_a.depth_max - scene_a.depth_min) * 0.5 ** it)
    nr = np.float32(0.5 * 0.5 ** it)
    ctx = scene_a.oracle_ctx(ref, srcs, k, "fast")
    d, n, c = g["depth"], g["normal"], g["cost"]
    od, on, oc = d, n, c
    for s in range(2):
        draw = 1 + it * 2 + s
        d, n, c = feng_a.refine_step(ref, srcs, k, d, n, c, seed, ref, draw, dr, nr, scene_a.depth_min, scene_a.depth_max)
        u, nz = oracle.rng_fill(seed, ref, draw, scene_a.H * scene_a.W)
        od, on, oc = ctx.refine_step(od, on, oc, u, nz, dr, nr, scene_a.depth_min, scene_a.depth_max)
        _eq(d, od, f"fast refine it{it} s{s} depth")
        _eq(c, oc, f"fast refine it{it} s{s} cost")
        _eq(n, on, f"fast refine it{it} s{s} normal")
    assert np.mean(d == g[f"depth_it{it}"]) >= 0.995              # reference golden


# ------------------------------------------------------------------ end to end ----
def _e2e_golden(depth, conf, want_d, want_c, what):
    rel = np.abs(depth - want_d) / want_d
    assert np.mean(rel <= 1e-3) >= E2E_MIN_FRACTION, f"{what}: {np.mean(rel <= 1e-3):.4f} within 1e-3"
    hg = np.bincount(conf.astype(int).ravel(), minlength=5) / conf.size
    hw = np.bincount(want_c.astype(int).ravel(), minlength=5) / want_c.size
    assert np.abs(hg - hw).max() < CONF_HIST_TOL, what


def test_patchmatch_bit_exact_golden_and_chamfer(feng_b, scene_b, amvs_mod):
    """g06 in fast mode: bit-exact against the oracle's fast mode, the reference's maps within the
    end-to-end tolerance, and the fused cloud within CHAMFER_TOL of the cloud the reference fused
    from its own maps (g14), both through the host geometry and through amvs_fuse_filter."""
    from amvs.core.mvs_patchmatch import DepthNormalMap, PatchMatchMVS
    from amvs.engine import make_pm_params
    g, g14 = load_golden("g06_patchmatch_e2e"), load_golden("g14_fused_cloud")
    refs = [int(r) for r in g["refs"]]
    srcs = [list(g[f"srcs_{r}"]) for r in refs]
    p = make_pm_params(int(g["patch"]), int(g["iters"]), int(g["samples"]), scene_b.depth_min, scene_b.depth_max)
    depth, normal, conf = feng_b.patchmatch(refs, srcs, p, int(g["seed"]))
    maps, proc = {}, {}
    for i, r in enumerate(refs):
        od, on, oc = scene_b.oracle_ctx(r, srcs[i], int(g["patch"]), "fast").patchmatch(
            int(g["iters"]), int(g["samples"]), scene_b.depth_min, scene_b.depth_max, int(g["seed"]), r)
        _eq(depth[i], od, f"fast view {r} depth")
        _eq(conf[i], oc, f"fast view {r} confidence")
        _eq(normal[i], on, f"fast view {r} normal")
        _e2e_golden(depth[i], conf[i], g[f"depth_{r}"], g[f"confidence_{r}"], f"view {r}")
        maps[r] = DepthNormalMap(depth=depth[i], normal=normal[i], confidence=conf[i])
        proc[r] = {"color": scene_b.colors[r]}
    for mv in (2, 3):
        pm = PatchMatchMVS(amvs_mod.Camera(K=scene_b.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7, min_views=mv)
        pts, cols = pm._fuse_depth_maps(maps, proc, scene_b.poses())
        fpts, _ = pm._filter_points(pts, cols)
        assert chamfer(pts, g14[f"points_mv{mv}"]) < CHAMFER_TOL
        assert chamfer(fpts, g14[f"f_points_mv{mv}"]) < CHAMFER_TOL
        dpts, _, raw = feng_b.fuse_filter(depth, conf, np.stack([scene_b.colors[r] for r in refs]),
                                          np.linalg.inv(scene_b.K), [(scene_b.R[r], scene_b.t[r]) for r in refs], mv, True)
        assert raw == len(pts) and np.array_equal(dpts, fpts)
        assert chamfer(dpts, g14[f"f_points_mv{mv}"]) < CHAMFER_TOL


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_patchmatch_baseline_schedule_golden(scene_d, mode):
    """g17: the BASELINE schedule (8 iterations x 8 samples, 7x7, S=4) captured from the reference:
    both modes bit-exact against their oracle mode and within the end-to-end tolerance."""
    from amvs.engine import make_pm_params
    g = load_golden("g17_patchmatch_long")
    r, srcs = int(g["ref"]), list(g["srcs"])
    eng = scene_d.engine(mode)
    try:
        p = make_pm_params(7, 8, 8, scene_d.depth_min, scene_d.depth_max)
        depth, normal, conf = eng.patchmatch([r], [srcs], p, int(g["seed"]))
    finally:
        eng.close()
    od, on, oc = scene_d.oracle_ctx(r, srcs, 7, mode).patchmatch(8, 8, scene_d.depth_min, scene_d.depth_max,
                                                                int(g["seed"]), r)
    _eq(depth[0], od, f"{mode} depth")
    _eq(conf[0], oc, f"{mode} confidence")
    _eq(normal[0], on, f"{mode} normal")
    _e2e_golden(depth[0], conf[0], g["depth"], g["confidence"], f"g17 {mode}")


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_plane_sweep_four_and_six_neighbours(scene_c, scene_d, mode):
    for name, sc in (("g11_plane_sweep", scene_c), ("g16_plane_sweep_s6", scene_d)):
        g = load_golden(name)
        ref, nbrs, k = int(g["ref"]), list(g["nbrs"]), int(g["patch"])
        depths = g["depths"].astype(np.float32)
        eng = sc.engine(mode)
        try:
            d, conf = eng.plane_sweep(ref, nbrs, depths, k, float(g["thresh"]))
            eng.set_sweep_tuning(tile_rows=5, planes_per_wave=3)          # several strips, several plane chunks
            d2, conf2 = eng.plane_sweep(ref, nbrs, depths, k, float(g["thresh"]))
        finally:
            eng.close()
        od, oc = sc.oracle_ctx(ref, nbrs, k, mode).plane_sweep(depths, float(g["thresh"]))
        _eq(d, od, f"{name} {mode} depth")
        _eq(conf, oc, f"{name} {mode} confidence")
        _eq(d2, od, f"{name} {mode} depth (chunked)")
        _eq(conf2, oc, f"{name} {mode} confidence (chunked)")
        assert np.mean(conf == g["confidence"]) > 0.995 and np.mean(d == g["depth_map"]) > 0.99
    # thresholds <= 0 take the division form of the vote in fast mode (the squared comparison needs t > 0)
    g = load_golden("g16_plane_sweep_s6")
    ref, nbrs, depths = int(g["ref"]), list(g["nbrs"]), g["depths"].astype(np.float32)
    eng = scene_d.engine(mode)
    try:
        for thresh in (0.0, -0.3, 0.35):
            d, conf = eng.plane_sweep(ref, nbrs, depths, 5, thresh)
            od, oc = scene_d.oracle_ctx(ref, nbrs, 5, mode).plane_sweep(depths, thresh)
            _eq(d, od, f"{mode} thresh {thresh} depth")
            _eq(conf, oc, f"{mode} thresh {thresh} confidence")
    finally:
        eng.close()


@pytest.mark.parametrize("shape,nviews,k,S", [((9, 11), 3, 3, 2), ((5, 70), 4, 5, 3), ((66, 7), 4, 7, 3),
                                               ((40, 90), 7, 11, 6), ((30, 64), 7, 9, 5), ((24, 58), 7, 5, 6),
                                               ((33, 59), 5, 7, 4), ((70, 117), 5, 7, 4)])
def test_tiny_ragged_and_wide_source_sets_bit_exact(amvs_mod, shape, nviews, k, S):
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    from oracle import oracle
    H, W = shape
    sc = make_scene(nviews, H, W, seed=H * 7 + W)
    grays = [(np.round(g * 255.0).astype(np.uint8)).astype(np.float32) / np.float32(255.0) for g in sc.grays]
    K = sc.camera.K.astype(np.float32)
    ref = nviews // 2
    srcs = [i for i in range(nviews) if i != ref][:S]
    with amvs_mod.Engine(H, W, nviews, K, mode="fast") as eng:
        for i in range(nviews):
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        assert eng.sampling_mode() == "u8-pairs" and eng.mode() == "fast"
        depth, normal, conf = eng.patchmatch([ref], [srcs], make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max), 3)
        ctx = oracle.ViewContext(K, grays[ref], sc.poses[ref].R, sc.poses[ref].t, [grays[i] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode="fast")
        od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 3, ref)
        _eq(depth[0], od, "depth")
        _eq(conf[0], oc, "confidence")
        _eq(normal[0], on, "normal")
        if k in (5, 7):
            depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, 9)).astype(np.float32)
            d, c = eng.plane_sweep(ref, srcs, depths, k, 0.8)
            wd, wc = ctx.plane_sweep(depths, 0.8)
            _eq(d, wd, "sweep depth")
            _eq(c, wc, "sweep confidence")


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_wide_baselines_bit_exact(amvs_mod, seed):
    """Wide baselines and a depth range that throws many projections far outside the sources."""
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
    with amvs_mod.Engine(H, W, nviews, K, mode="fast") as eng:
        for i in range(nviews):
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        depth, normal, conf = eng.patchmatch([ref], [srcs], make_pm_params(k, 2, 3, dmin, dmax), 40 + seed)
        ctx = oracle.ViewContext(K, grays[ref], sc.poses[ref].R, sc.poses[ref].t, [grays[i] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode="fast")
        od, on, oc = ctx.patchmatch(2, 3, dmin, dmax, 40 + seed, ref)
        tag = f"seed {seed}: {H}x{W} k{k} S{S}"
        _eq(depth[0], od, tag + " depth")
        _eq(conf[0], oc, tag + " confidence")
        _eq(normal[0], on, tag + " normal")


def test_mode_selection_and_errors(scene_b, amvs_mod):
    """amvs_pm_params.mode overrides the engine's mode; fast mode refuses images that are not 8-bit."""
    from amvs._lib import AmvsError
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    eng = scene_b.engine("exact")
    try:
        kw = dict(patch_size=7, num_iterations=1, num_samples=1, depth_min=scene_b.depth_min, depth_max=scene_b.depth_max)
        exact = eng.patchmatch([1], [[0, 2, 3, 4]], make_pm_params(**kw), 1)
        fast = eng.patchmatch([1], [[0, 2, 3, 4]], make_pm_params(mode="fast", **kw), 1)
        eng.set_mode("fast")
        assert eng.mode() == "fast"
        fast2 = eng.patchmatch([1], [[0, 2, 3, 4]], make_pm_params(**kw), 1)
        exact2 = eng.patchmatch([1], [[0, 2, 3, 4]], make_pm_params(mode="exact", **kw), 1)
    finally:
        eng.close()
    _eq(fast[0], fast2[0], "fast by params == fast by engine")
    _eq(exact[0], exact2[0], "exact by engine == exact by params")
    # the two arithmetics pick the same hypotheses almost everywhere (here: everywhere) ...
    assert np.mean(np.abs(fast[0] - exact[0]) <= 1e-3 * exact[0]) > 0.99
    # ... although their costs differ in the last bits
    d = scene_b.gt_depth[1]
    ce, cf = [], []
    for m, acc in (("exact", ce), ("fast", cf)):
        e3 = scene_b.engine(m)
        try:
            acc.append(e3.eval_cost(1, [0, 2, 3, 4], 7, d))
        finally:
            e3.close()
    fin = np.isfinite(ce[0]) & np.isfinite(cf[0])
    # (a knife-edge validity flip changes a cost by a source's share: compared as a quantile)
    assert not np.array_equal(ce[0], cf[0]) and np.quantile(np.abs(ce[0] - cf[0])[fin], 0.999) < 1e-4
    sc = make_scene(3, 40, 70, seed=2)                               # rendered floats: not 8-bit exact
    with amvs_mod.Engine(40, 70, 3, sc.camera.K.astype(np.float32), mode="fast") as e2:
        for i in range(3):
            e2.set_view(i, sc.grays[i], sc.poses[i].R, sc.poses[i].t)
        with pytest.raises(AmvsError, match="8-bit"):
            e2.eval_cost(1, [0, 2], 7, sc.depths[1])
