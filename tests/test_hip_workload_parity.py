"""BASELINE configs 4 and 5 as WORKLOADS (not only their kernel shapes), and the strip orders.

    config 4   32 views of 1920x1080, the full 8 x (2 + 8) schedule, swept through
               PatchMatchMVS._sweep_resident by TWO ranks (gloo, both on cuda:0) in launches of 8 and
               of 4 views -- the shard shapes of a 4- and an 8-GPU split -- and all-gathered: the
               gathered maps of every rank equal the single-process maps, two of which are compared
               bit for bit with the CPU oracle (reference loop: mvs_patchmatch.py:104-123, :287-308)
    config 5   64 views of 3840x2160 through the drop-in class (device image preparation, sweep in
               launches of 16 views, 1 x 1 schedule), fusion + filter of all 64 maps on the device equal
               to the NumPy fusion (mvs_patchmatch.py:536-588) of the downloaded maps, two views vs the
               oracle
    strips     band-major == view-major == split schedule, bit for bit
"""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

C4 = dict(n=32, H=1080, W=1920, iters=8, samples=8, seed=42, scene_seed=1234)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _eq(a, b, what):
    a = np.asarray(a)
    b = np.asarray(b)
    same = (a == b) | (np.isnan(a) & np.isnan(b))
    assert same.all(), f"{what}: {int((~same).sum())} of {same.size} elements differ " \
                       f"(first at {np.argwhere(~same)[0]}: {a[~same][0]!r} vs {b[~same][0]!r})"


def _digest(t):
    """sha1 of a device tensor's bytes (one row of a resident map)."""
    return hashlib.sha1(t.contiguous().cpu().numpy().tobytes()).hexdigest()


def _config4_sweep(views_per_batch, keep_rows=(), mode="fast"):
    """The config-4 scene through PatchMatchMVS._sweep_resident (sharded + gathered when a process
    group is initialised).  Returns ({view: (sha1 depth, sha1 normal, sha1 confidence)}, {view: maps
    of the rows asked for}, the prepared scene)."""
    import torch

    import amvs
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    from amvs.synthetic import make_scene
    c = C4
    sc = make_scene(c["n"], c["H"], c["W"], seed=c["scene_seed"], device="cuda")
    sc.grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
                for g in sc.grays]
    ids = sorted(sc.poses)
    pm = PatchMatchMVS(amvs.Camera(K=sc.camera.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                       num_iterations=c["iters"], num_samples=c["samples"], min_views=3, seed=c["seed"],
                       views_per_batch=views_per_batch, device=0, mode=mode, device_prep=False)
    pm.depth_min, pm.depth_max = sc.depth_min, sc.depth_max
    proc = {i: {"gray": sc.grays[i], "color": sc.colors[i], "shape": (c["H"], c["W"])} for i in ids}
    jobs = [(r, pm._select_source_views(r, ids, sc.poses, k=4)) for r in ids]
    res = pm._sweep_resident(torch, jobs, proc, sc.poses, ids)
    torch.cuda.synchronize()
    assert res.ref_ids == ids
    digests = {r: (_digest(res.depth[i]), _digest(res.normal[i]), _digest(res.confidence[i]))
               for i, r in enumerate(res.ref_ids)}
    H, W = c["H"], c["W"]
    kept = {r: (res.depth[r].cpu().numpy().reshape(H, W), res.normal[r].cpu().numpy().reshape(H, W, 3),
                res.confidence[r].cpu().numpy().reshape(H, W)) for r in keep_rows}
    pm._engine.close()
    return digests, kept, (sc, jobs)


def _config4_worker(rank, world, port, q, views_per_batch, mode):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        digests, _, _ = _config4_sweep(views_per_batch, mode=mode)
        q.put((rank, digests))
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module", params=["fast", "exact"])
def config4_single(request):
    """(digests, kept maps, scene, arithmetic mode) of the single-process sweep -- in the fast arithmetic the bench
    times and in the exact one the classes default to (round 4)."""
    return _config4_sweep(16, keep_rows=(5, 26), mode=request.param) + (request.param,)


@pytest.mark.timeout(900)
def test_config4_single_process_views_match_the_oracle(config4_single):
    from oracle import oracle
    oracle.set_threads(16)
    _, kept, (sc, jobs), mode = config4_single
    c = C4
    for r, (d, n, cf) in kept.items():
        srcs = jobs[r][1]
        ctx = oracle.ViewContext(sc.camera.K.astype(np.float32), sc.grays[r], sc.poses[r].R, sc.poses[r].t,
                                 [sc.grays[i] for i in srcs], [sc.poses[i].R for i in srcs],
                                 [sc.poses[i].t for i in srcs], 7, mode=mode)
        od, on, oc = ctx.patchmatch(c["iters"], c["samples"], sc.depth_min, sc.depth_max, c["seed"], r)
        _eq(d, od, f"config 4 ({mode}) view {r} depth")
        _eq(cf, oc, f"config 4 ({mode}) view {r} confidence")
        _eq(n, on, f"config 4 ({mode}) view {r} normal")


@pytest.mark.timeout(900)
@pytest.mark.parametrize("views_per_batch", [8, 4])
def test_config4_two_ranks_rank_shaped_shards_gathered(config4_single, views_per_batch):
    """Two gloo ranks on cuda:0, 16 views each, swept in launches of 8 (the 4-GPU shard) or 4 views (the
    8-GPU shard) and all-gathered: every rank ends with the single-process maps of all 32 views."""
    import torch.multiprocessing as mp
    single, _, _, mode = config4_single
    if mode == "exact" and views_per_batch != 8:
        pytest.skip("the exact arithmetic runs the 4-GPU-shaped shard only (time)")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_config4_worker, args=(r, 2, port, q, views_per_batch, mode)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, digests in results:
        assert sorted(digests) == sorted(single)
        bad = [r for r in single if digests[r] != single[r]]
        assert not bad, f"rank {rank}, {views_per_batch} views per launch ({mode}): gathered maps of views {bad} differ"


@pytest.mark.timeout(1200)
def test_config5_64x4k_workload_fusion_on_device():
    """BASELINE config 5 on one GPU: 64 views of 3840x2160 through the class's own pipeline -- 8-bit BGR
    uploads prepared on the device, sweep in launches of 16 views, maps resident, fusion + filter of
    all 64 maps on the device -- against the NumPy fusion of the downloaded maps and, for two views,
    the oracle."""
    import torch

    import amvs
    from amvs.core.imageprep import prepare_view
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    from amvs.synthetic import make_scene
    from oracle import oracle
    n, H, W = 64, 2160, 3840
    sc = make_scene(n, H, W, seed=4, device="cuda")
    sc.grays = sc.depths = None                                    # (2 x 2.1 GB of host arrays nobody reads here)
    ids = sorted(sc.poses)
    pm = PatchMatchMVS(amvs.Camera(K=sc.camera.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                       num_iterations=1, num_samples=1, min_views=3, seed=7, views_per_batch=16, device=0,
                       mode="fast", device_prep=True)
    pm.depth_min, pm.depth_max = sc.depth_min, sc.depth_max
    proc = pm._prepare_images_device(sc.images(), ids, sc.poses)
    jobs = [(r, pm._select_source_views(r, ids, sc.poses, k=4)) for r in ids]
    res = pm._sweep_resident(torch, jobs, proc, sc.poses, ids)
    assert pm._engine.last_views_per_launch() == 16                # 4K: the whole 16-view batch per launch
    pts, cols, raw = pm._fuse_filter_resident(res, proc, sc.poses)
    assert raw > 1000 and len(pts) > 1000, (raw, len(pts))
    host = res.to_host()
    hp, hc = pm._fuse_depth_maps(host, proc, sc.poses)
    assert len(hp) == raw
    hp, hc = pm._filter_points(hp, hc)
    assert np.array_equal(pts, hp), "device fusion + filter of the 64 maps differs from the NumPy fusion"
    assert np.array_equal(cols, hc)
    oracle.set_threads(16)
    for r in (9, 40):
        srcs = jobs[r][1]
        g = {i: prepare_view(sc.colors[i], 1.0)["gray"] for i in [r] + srcs}
        ctx = oracle.ViewContext(sc.camera.K.astype(np.float32), g[r], sc.poses[r].R, sc.poses[r].t,
                                 [g[i] for i in srcs], [sc.poses[i].R for i in srcs],
                                 [sc.poses[i].t for i in srcs], 7, mode="fast")
        od, on, oc = ctx.patchmatch(1, 1, sc.depth_min, sc.depth_max, 7, r)
        _eq(host[r].depth, od, f"config 5 view {r} depth")
        _eq(host[r].confidence, oc, f"config 5 view {r} confidence")
        _eq(host[r].normal, on, f"config 5 view {r} normal")
    pm._engine.close()


@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_strip_orders_give_identical_maps(mode):
    """amvs_pm_params.schedule: band-major strips (every XCD walks one band of ALL views, strip height
    from pick_band_rows) and -- fast arithmetic -- the split schedule (sampling kernel + window kernel
    on two streams) return the view-major maps bit for bit; 6 views of 540x960, 2 x (2 + 3) steps, so
    that bands of several views share an XCD and strips end inside the image."""
    import torch

    import amvs
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    n, H, W = 6, 540, 960
    sc = make_scene(n, H, W, seed=21, device="cuda" if torch.cuda.is_available() else "cpu")
    grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
             for g in sc.grays]
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    maps, rows = {}, {}
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        for schedule in ["view-major", "band-major", "paired"] + (["split"] if mode == "fast" else []):
            p = make_pm_params(7, 2, 3, sc.depth_min, sc.depth_max, schedule=schedule)
            maps[schedule] = eng.patchmatch(ids, sources, p, 11)
            rows[schedule] = eng.last_tile_rows()
        # paired bands (both arithmetic modes have their own PAIR kernel) with an even band count whose last band
        # is short (540 = 33 x 16 + 12), an odd one (27 bands of 20 rows: the last band has no partner) and
        # single-band strips (one band of 540)
        for tr in (16, 20, 540):
            p = make_pm_params(7, 2, 3, sc.depth_min, sc.depth_max, schedule="paired", tile_rows=tr)
            maps[f"paired/{tr}"] = eng.patchmatch(ids, sources, p, 11)
    assert rows["band-major"] != rows["view-major"], rows          # a different launch shape was really used
    for schedule in maps:
        for a, b, what in zip(maps["view-major"], maps[schedule], ("depth", "normal", "confidence")):
            _eq(b, a, f"{mode} {schedule} {what}")


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("k", [9, 11, 13])
def test_paired_bands_for_large_patches(mode, k):
    """Round 4: paired bands for 9x9 and 11x11 patches -- the partner band's rows of the LDS-resident sources are
    read from the partner's ring itself, the register-resident ones through exchange rows.  Even band counts with
    a short last band (300 = 18 x 16 + 12), an odd count (15 bands of 20), bands shorter than the halo (4 rows:
    the partner's "own rows next to the boundary" reach into its far halo), one band per view -- all equal to the
    classic strips bit for bit, which the other parity tests pin to the oracle.  (13 x 13: compiled, not paired --
    a request for the paired schedule runs the classic strips; same maps, and the oracle comparison below covers the
    compiled 13 x 13 kernels at several strip heights on a multi-strip image.)"""
    import torch

    import amvs
    from amvs.engine import make_pm_params
    from amvs.synthetic import make_scene
    n, H, W = 5, 300, 500
    sc = make_scene(n, H, W, seed=31, device="cuda" if torch.cuda.is_available() else "cpu")
    grays = [(np.round(g * 255.0).clip(0, 255).astype(np.uint8)).astype(np.float32) / np.float32(255.0)
             for g in sc.grays]
    ids = sorted(sc.poses)
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    sources = [pm._select_source_views(r, ids, sc.poses, k=4) for r in ids]
    with amvs.Engine(H, W, n, sc.camera.K.astype(np.float32), mode=mode) as eng:
        for i in ids:
            eng.set_view(i, grays[i], sc.poses[i].R, sc.poses[i].t)
        want = eng.patchmatch(ids, sources, make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max, schedule="view-major"), 13)
        for tr in (0, 16, 20, 4, 300):
            got = eng.patchmatch(ids, sources, make_pm_params(k, 2, 2, sc.depth_min, sc.depth_max, schedule="paired", tile_rows=tr), 13)
            for a, b, what in zip(want, got, ("depth", "normal", "confidence")):
                _eq(b, a, f"{mode} k{k} paired/{tr} {what}")
    # and against the oracle directly (view 2)
    from oracle import oracle
    srcs = sources[2]
    ctx = oracle.ViewContext(sc.camera.K.astype(np.float32), grays[2], sc.poses[2].R, sc.poses[2].t, [grays[i] for i in srcs],
                             [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], k, mode=mode)
    od, on, oc = ctx.patchmatch(2, 2, sc.depth_min, sc.depth_max, 13, 2)
    _eq(want[0][2], od, f"{mode} k{k} depth vs the oracle")
    _eq(want[1][2], on, f"{mode} k{k} normal vs the oracle")
    _eq(want[2][2], oc, f"{mode} k{k} confidence vs the oracle")


@pytest.mark.parametrize("mode", ["fast", "exact"])
@pytest.mark.parametrize("entry", ["host", "device"])
def test_sweep_continued_one_iteration_per_call(scene_a, mode, entry):
    """amvs_pm_params.first_iteration: a sweep run as one call per iteration (the confidence pass only in
    the last) returns the maps of the single call bit for bit -- through the host-buffer entry point
    (which resolves the state in place between the calls) and the device one -- and a continuation that
    does not match the previous call is refused.  This is what bench.py --gather-per-iteration runs."""
    import torch

    import amvs
    from amvs.engine import make_pm_params
    sc = scene_a
    refs, srcs = [2, 1], [[1, 3, 0, 4], [0, 2, 3, 4]]
    iters, samples = 3, 4

    def run(eng, **kw):
        p = make_pm_params(7, kw.pop("n", iters), samples, sc.depth_min, sc.depth_max, **kw)
        if entry == "host":
            return eng.patchmatch(refs, srcs, p, 9)
        dev = torch.device("cuda", 0)
        out = [torch.zeros((2, sc.H * sc.W * k), dtype=torch.float32, device=dev) for k in (1, 3, 1)]
        torch.cuda.synchronize()
        eng.patchmatch_device(refs, srcs, p, 9, *[t.data_ptr() for t in out])
        eng.sync()
        return tuple(t.cpu().numpy() for t in out)

    with sc.engine(mode) as eng:
        whole = run(eng)
        for it in range(iters):
            last = run(eng, n=1, first_iteration=it, confidence=it == iters - 1)
        for a, b, what in zip(whole, last, ("depth", "normal", "confidence")):
            _eq(b.reshape(a.shape), a, f"{mode} {entry} {what} after {iters} one-iteration calls")
        with pytest.raises(amvs.AmvsError):
            run(eng, n=1, first_iteration=1)                     # 3 iterations were run, not 1
        with pytest.raises(amvs.AmvsError):
            eng.patchmatch([2], [[1, 3, 0, 4]], make_pm_params(7, 1, samples, sc.depth_min, sc.depth_max, first_iteration=3), 9)


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_cli_default_reconstruct_equals_an_oracle_backed_run(mode, capsys):
    """What run_reconstruction.py constructs (run_reconstruction.py:131-136, mvs_patchmatch.py:43-50):
    PatchMatchMVS(camera, scale=0.25, num_iterations=3, min_views=3) -- patch 11, 3 x (2 + 8), depth range from the
    camera spread (no sparse points) -- through reconstruct() end to end on 8-bit BGR inputs: device image
    preparation, sweep, device fusion + filter.  The same run with every device step replaced: the host restatement
    of the image preparation, the CPU oracle for every view's sweep, the NumPy fusion / filter of
    core/mvs_patchmatch.py.  The two clouds must be identical, point for point."""
    import torch

    import amvs
    from amvs.core.imageprep import prepare_views
    from amvs.synthetic import make_scene
    from oracle import oracle
    oracle.set_threads(16)
    n, h, w = 6, 432, 576                                   # processed at 108 x 144
    sc = make_scene(n, h, w, seed=77, device="cuda" if torch.cuda.is_available() else "cpu")
    cam = amvs.Camera(K=sc.camera.K.copy(), dist=np.zeros(5))
    pm = amvs.PatchMatchMVS(cam, scale=0.25, num_iterations=3, min_views=3, seed=5, mode=mode)
    assert (pm.patch_size, pm.num_samples) == (11, 8)
    pts, cols = pm.reconstruct(sc.images(), sc.poses)
    capsys.readouterr()
    # ---- the same pipeline without the device ----
    ids = sorted(sc.poses)
    prepared = prepare_views([sc.colors[i] for i in ids], 0.25)
    proc = dict(zip(ids, prepared))
    K32 = pm.K_scaled.astype(np.float32)
    maps = {}
    for slot, r in enumerate(ids):
        srcs = pm._select_source_views(r, ids, sc.poses, k=4)
        ctx = oracle.ViewContext(K32, proc[r]["gray"], sc.poses[r].R, sc.poses[r].t, [proc[i]["gray"] for i in srcs],
                                 [sc.poses[i].R for i in srcs], [sc.poses[i].t for i in srcs], 11, mode=mode)
        d, nr, cf = ctx.patchmatch(3, 8, pm.depth_min, pm.depth_max, 5, slot)
        maps[r] = amvs.DepthNormalMap(depth=d, normal=nr, confidence=cf)
        ctx.close()
    want_p, want_c = pm._fuse_depth_maps(maps, proc, sc.poses)
    assert len(want_p) > 0, "the scene must fuse some points for the comparison to mean anything"
    want_p, want_c = pm._filter_points(want_p, want_c)
    assert pts.shape == want_p.shape and np.array_equal(pts, want_p), f"{mode}: clouds differ ({len(pts)} vs {len(want_p)} points)"
    assert np.array_equal(cols, want_c)


def test_state_invalidation_flags_and_stale_step_times(scene_a):
    """ADVICE (round 3).  (1) The host plane sweep writes its maps into slot 0 of the PatchMatch state: a
    continuation after it is refused.  (2) AMVS_PM_NO_CONFIDENCE leaves the caller's confidence array
    untouched (host entry point, fused and split schedules).  (3) amvs_get_step_times after a call that
    records no per-launch events (plane sweep, split schedule) returns none instead of stale intervals."""
    import amvs
    from amvs.engine import make_pm_params
    sc = scene_a
    refs, srcs = [2], [[1, 3, 0, 4]]
    depths = (1.0 / np.linspace(1 / sc.depth_max, 1 / sc.depth_min, 8)).astype(np.float32)
    with sc.engine("fast") as eng:
        eng.set_step_timing(True)
        p1 = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max, confidence=False)
        d, n, c = eng.patchmatch(refs, srcs, p1, 9)
        assert len(eng.step_times()) == 4                       # 2 propagation + 2 refinement launches
        # (2) the confidence array comes back as numpy allocated it: fill it through the binding by hand
        import ctypes as C
        from amvs._lib import f32p, i32p
        conf = np.full((1, sc.H, sc.W), -7.0, np.float32)
        dd = np.empty((1, sc.H, sc.W), np.float32)
        nn = np.empty((1, sc.H, sc.W, 3), np.float32)
        ref = np.asarray(refs, np.int32)
        src = np.asarray(srcs, np.int32)
        for schedule in ("auto", "split"):
            p = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max, confidence=False, schedule=schedule)
            rc = eng._lib.amvs_patchmatch(eng._h, 1, ref.ctypes.data_as(i32p), src.ctypes.data_as(i32p), 4, C.byref(p), 9,
                                          dd.ctypes.data_as(f32p), nn.ctypes.data_as(f32p), conf.ctypes.data_as(f32p))
            assert rc == 0 and np.all(conf == -7.0), schedule
            assert np.array_equal(dd[0], d[0]), schedule
        assert len(eng.step_times()) == 0                       # (3) the split schedule records none
        eng.patchmatch(refs, srcs, p1, 9)
        assert len(eng.step_times()) == 4
        # (1), (3): the plane sweep lands in slot 0 and records no step events
        eng.plane_sweep(2, [1, 3, 0, 4], depths, 5, 0.8)
        assert len(eng.step_times()) == 0
        with pytest.raises(amvs.AmvsError):
            eng.patchmatch(refs, srcs, make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max, first_iteration=1), 9)


def test_native_rccl_entry_points_single_rank():
    """amvs_comm_unique_id / amvs_comm_init / amvs_allgather_maps / amvs_comm_destroy (include/amvs.h): the
    native exchange of a C-ABI consumer without torch.distributed, RCCL resolved with dlopen.  One GPU
    here, so a one-rank communicator: the all-gather, enqueued on the engine's stream behind a sweep,
    must deliver this rank's maps -- out of place and in place.  (Two ranks need two GPUs.)"""
    import torch

    import amvs
    from amvs.engine import make_pm_params
    from conftest import GoldenScene
    sc = GoldenScene("scene_a")
    dev = torch.device("cuda", 0)
    hw = sc.H * sc.W
    with sc.engine("fast") as eng:
        eng.comm_init(0, 1, amvs.Engine.comm_unique_id())
        d = torch.zeros((1, hw), dtype=torch.float32, device=dev)
        n = torch.zeros((1, 3 * hw), dtype=torch.float32, device=dev)
        c = torch.zeros((1, hw), dtype=torch.float32, device=dev)
        full = torch.full((1, hw), -1.0, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        p = make_pm_params(7, 1, 2, sc.depth_min, sc.depth_max)
        eng.patchmatch_device([2], [[1, 3, 0, 4]], p, 5, d.data_ptr(), n.data_ptr(), c.data_ptr())
        eng.allgather_maps(d.data_ptr(), full.data_ptr(), hw)          # stream-ordered behind the sweep
        eng.allgather_maps(n.data_ptr(), n.data_ptr(), 3 * hw)         # in place
        eng.sync()
        assert torch.equal(full, d) and float(d.min()) >= sc.depth_min - 1e-6
        want = eng.patchmatch([2], [[1, 3, 0, 4]], p, 5)
        assert np.array_equal(n.cpu().numpy().reshape(sc.H, sc.W, 3), want[1][0])
        eng.comm_destroy()
        with pytest.raises(amvs.AmvsError):
            eng.allgather_maps(d.data_ptr(), full.data_ptr(), hw)      # no communicator any more


def test_reused_context_gives_a_fresh_objects_results(capsys):
    """Round 4: both classes keep their device context between reconstruct calls on the same problem size and upload
    the next call's views into it.  A second call with OTHER images and poses of the same size must return exactly
    what a fresh object returns for them (no state of the first scene survives: reference statistics, packed maps,
    resident clouds, continuation state), and a call with another size must get a new context."""
    import amvs as amvs_mod
    from amvs.synthetic import make_scene
    a = make_scene(5, 96, 128, seed=3)
    b = make_scene(5, 96, 128, seed=9, arc_step_deg=14.0)
    c = make_scene(5, 80, 112, seed=4)
    cam = amvs_mod.Camera(K=a.camera.K.copy(), dist=np.zeros(5))
    cam_c = amvs_mod.Camera(K=c.camera.K.copy(), dist=np.zeros(5))

    def mvs(camera):
        return amvs_mod.PatchMatchMVS(camera, scale=1.0, patch_size=7, num_iterations=2, num_samples=3, min_views=2, seed=5,
                                      device_prep=True)

    def stereo(camera):
        return amvs_mod.DenseStereoReconstructor(camera, scale=1.0, num_depths=24, min_views=2, device_prep=True)
    for make in (mvs, stereo):
        kept = make(cam)
        kept.reconstruct(a.images(), a.poses)
        first_engine = kept._engine
        got = kept.reconstruct(b.images(), b.poses)
        assert kept._engine is first_engine                      # reused
        want = make(cam).reconstruct(b.images(), b.poses)
        assert len(want[0]) > 100
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        again = kept.reconstruct(a.images(), a.poses)
        fresh_a = make(cam).reconstruct(a.images(), a.poses)
        assert np.array_equal(again[0], fresh_a[0]) and np.array_equal(again[1], fresh_a[1])
        kept.camera, kept.K_scaled = cam_c, cam_c.K.copy()       # another size (and intrinsics): a new context
        other = kept.reconstruct(c.images(), c.poses)
        assert kept._engine is not first_engine
        want_c = make(cam_c).reconstruct(c.images(), c.poses)
        assert np.array_equal(other[0], want_c[0]) and np.array_equal(other[1], want_c[1])
    capsys.readouterr()
