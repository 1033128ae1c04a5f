"""The N>1 path on CPU: two gloo ranks shard the reference views, compute their block and
all-gather the per-view maps (amvs.parallel); the result must equal the single-process one.
Per-view compute is injected (the CPU oracle stands in for the HIP engine, which needs a GPU)."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_backend(scene, patch, iters, samples, seed, mode="exact"):
    """Sweep backend on the CPU oracle, fed with the same prepared gray maps the engine gets
    (BGR -> gray of the colour images, as reconstruct() does)."""
    from oracle import oracle
    from amvs.core.imageprep import prepare_view
    grays = [prepare_view(c, 1.0)["gray"] for c in scene.colors]

    def run(jobs):
        d, n, c = [], [], []
        for ref, srcs in jobs:
            ctx = oracle.ViewContext(scene.K32(), grays[ref], scene.R[ref], scene.t[ref],
                                     [grays[i] for i in srcs], [scene.R[i] for i in srcs],
                                     [scene.t[i] for i in srcs], patch, mode=mode)
            a, b, e = ctx.patchmatch(iters, samples, scene.depth_min, scene.depth_max, seed, ref)
            d.append(a); n.append(b); c.append(e)
        return np.stack(d), np.stack(n), np.stack(c)
    return run


def _reconstruct(scene, world_tag, use_engine=False, mode="exact"):
    """PatchMatchMVS.reconstruct; without a GPU the device sweep is replaced by the oracle backend
    (in the same arithmetic mode)."""
    import amvs
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    pm = PatchMatchMVS(amvs.Camera(K=scene.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                       num_iterations=1, num_samples=2, min_views=2, seed=5, views_per_batch=2, device=0, mode=mode,
                       device_prep=use_engine)
    if not use_engine:
        backend = _oracle_backend(scene, 7, 1, 2, 5, mode)
        pm._ensure_engine = lambda images, poses, indices: setattr(pm, "_slot", {i: i for i in indices})
        pm._run_batch = lambda eng, batch: backend(batch)
        pm.device_fusion = False        # host maps + NumPy fusion (bit-identical to the device path)
    # keep the synthetic scene's depth range (the reference would estimate it from sparse points)
    pm._estimate_depth_range = lambda poses, sparse: None
    pm.depth_min, pm.depth_max = scene.depth_min, scene.depth_max
    return pm.reconstruct([{"image": c} for c in scene.colors], scene.poses())


def _worker(rank, world, port, q, use_engine=False, mode="exact", backend="gloo"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      AMVS_ORACLE_THREADS="2")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import GoldenScene
    from amvs import parallel
    if backend == "nccl":
        # one rank per GPU over RCCL: bind the device before anything touches it
        os.environ["LOCAL_RANK"] = str(rank)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. raw collective: uneven shards (5 items over 2 ranks -> 3 + 2)
        mine = parallel.shard(5, rank, world)
        local = torch.tensor([[float(j), j * 10.0, j + 0.5] for j in mine], dtype=torch.float32).reshape(len(mine), 3)
        if backend == "nccl":
            local = local.cuda()
        full = parallel.allgather_packed(local, 5, 3).cpu()
        want = torch.tensor([[float(j), j * 10.0, j + 0.5] for j in range(5)], dtype=torch.float32)
        assert torch.equal(full, want), (rank, full)
        # 2. the sharded reconstruct path
        pts, cols = _reconstruct(GoldenScene("scene_a"), f"rank{rank}", use_engine, mode)
        q.put((rank, pts, cols))
    finally:
        dist.destroy_process_group()


def test_shard_is_a_partition():
    from amvs.parallel import shard, shard_sizes
    for n in (1, 5, 16, 17, 32):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard(n, r, world) for r in range(world)]
            assert sorted(sum(blocks, [])) == list(range(n))
            assert [len(b) for b in blocks] == shard_sizes(n, world)
            assert max(len(b) for b in blocks) == -(-n // world)


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process(scene_a):
    single_pts, single_cols = _reconstruct(scene_a, "single")
    assert single_pts.shape[0] > 0 and single_pts.shape[1] == 3
    for rank, pts, cols in _run_two_ranks(False):
        assert np.array_equal(pts, single_pts), f"rank {rank} cloud differs from the single-process cloud"
        assert np.array_equal(cols, single_cols)


def _run_two_ranks(use_engine, mode="exact", backend="gloo"):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")                    # fresh processes: nothing has touched a GPU yet
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_engine, mode, backend)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("mode", ["fast", "exact"])
def test_reconstruct_on_gpu_single_and_two_ranks(scene_a, mode):
    """The drop-in class end to end on the HIP engine: (1) its fused cloud equals the one obtained
    with the CPU oracle (same arithmetic mode) as sweep backend (the sweep is bit-exact, the host
    geometry is shared); (2) two ranks (gloo, both on cuda:0) sharding the views return the same
    cloud on every rank."""
    gpu_pts, gpu_cols = _reconstruct(scene_a, "gpu", use_engine=True, mode=mode)
    cpu_pts, cpu_cols = _reconstruct(scene_a, "oracle", use_engine=False, mode=mode)
    assert gpu_pts.shape[0] > 0
    assert np.array_equal(gpu_pts, cpu_pts) and np.array_equal(gpu_cols, cpu_cols)
    for rank, pts, cols in _run_two_ranks(True, mode):
        assert np.array_equal(pts, gpu_pts), f"rank {rank}"
        assert np.array_equal(cols, gpu_cols)


def _gpu_count():
    try:
        import torch
        return torch.cuda.device_count()          # does not initialise the GPU on this image
    except Exception:  # noqa: BLE001
        return 0


@pytest.mark.gpu
@pytest.mark.timeout(600)
@pytest.mark.skipif(_gpu_count() < 2, reason="the RCCL path needs two GPUs (one rank per GPU)")
def test_reconstruct_two_ranks_rccl_matches_single_process(scene_a):
    """One rank per GPU, backend nccl (= RCCL over xGMI): the device-resident all-gather of
    _sweep_resident and the fused cloud must equal the single-process result.  Skipped on the
    one-GPU test boxes; runs wherever two devices are visible.  Workers are fresh spawned processes
    that bind their device before any other GPU call."""
    single_pts, single_cols = _reconstruct(scene_a, "single", use_engine=True, mode="fast")
    for rank, pts, cols in _run_two_ranks(True, "fast", backend="nccl"):
        assert np.array_equal(pts, single_pts), f"rank {rank} cloud differs from the single-process cloud"
        assert np.array_equal(cols, single_cols)


def _rccl_one_rank_worker(port, q, mode):
    """A ONE-rank nccl (= RCCL) process group on cuda:0 with the multi-rank code path forced on: the
    row groups, the second stream, the event ordering and the RCCL all-gather calls of _sweep_resident
    execute for real (with a single peer); the stereo path's device gather likewise."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import GoldenScene
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        import amvs
        from amvs.core.mvs_patchmatch import PatchMatchMVS
        scene = GoldenScene("scene_a")
        pm = PatchMatchMVS(amvs.Camera(K=scene.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7, num_iterations=2,
                           num_samples=2, min_views=2, seed=5, views_per_batch=2, device=0, mode=mode, device_prep=True)
        pm.exercise_exchange = True
        pm._estimate_depth_range = lambda poses, sparse: None
        pm.depth_min, pm.depth_max = scene.depth_min, scene.depth_max
        q.put(pm.reconstruct([{"image": c} for c in scene.colors], scene.poses()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_reconstruct_rccl_code_path_on_a_one_rank_group(scene_a):
    """The nccl branches of _sweep_resident (direct device all-gathers, group by group, on the comm
    stream) executed on real RCCL -- with the only topology a one-GPU box offers, a one-rank group --
    return the cloud of the plain single-process run."""
    import amvs
    import torch.multiprocessing as mp
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    pm = PatchMatchMVS(amvs.Camera(K=scene_a.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7, num_iterations=2,
                       num_samples=2, min_views=2, seed=5, views_per_batch=2, device=0, mode="fast", device_prep=True)
    pm._estimate_depth_range = lambda poses, sparse: None
    pm.depth_min, pm.depth_max = scene_a.depth_min, scene_a.depth_max
    want_p, want_c = pm.reconstruct([{"image": c} for c in scene_a.colors], scene_a.poses())
    assert len(want_p) > 0
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_one_rank_worker, args=(_free_port(), q, "fast"))
    p.start()
    pts, cols = q.get(timeout=240)
    p.join(timeout=60)
    assert p.exitcode == 0
    assert np.array_equal(pts, want_p) and np.array_equal(cols, want_c)


def _stereo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank,) + _stereo_reconstruct())
    finally:
        dist.destroy_process_group()


def _stereo_reconstruct():
    from conftest import GoldenScene
    from amvs.core.dense_stereo import DenseStereoReconstructor
    import amvs
    sc = GoldenScene("scene_d")
    rec = DenseStereoReconstructor(amvs.Camera(K=sc.K.copy(), dist=np.zeros(5)), scale=1.0, num_depths=24, patch_size=5,
                                   min_views=2, device=0)
    return rec.reconstruct([{"image": c} for c in sc.colors], sc.poses(), max_pairs=30)


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_stereo_reconstruct_two_ranks_match_single_process():
    """DenseStereoReconstructor.reconstruct with the reference views sharded over two ranks (gloo,
    both on cuda:0; 7 views -> 4 + 3) returns the single-process cloud on every rank."""
    import torch.multiprocessing as mp
    single_p, single_c = _stereo_reconstruct()
    assert len(single_p) > 100
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stereo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, pts, cols in results:
        assert np.array_equal(pts, single_p) and np.array_equal(cols, single_c), f"rank {rank}"


def test_engine_cache_follows_the_poses(monkeypatch, scene_a):
    """ADVICE (round 1): the engine cache must not serve stale poses.  _patchmatch_cuda takes the
    poses on every call, as the reference does (mvs_patchmatch.py:225-257): changed poses, a
    different prepared dict or a different image size re-upload; identical inputs reuse the engine."""
    import amvs
    from amvs import engine as engine_mod
    from amvs.core.mvs_patchmatch import PatchMatchMVS

    uploads = []

    class FakeEngine:
        def __init__(self, H, W, n, K, device=0, mode="exact"):
            self.H, self.W = H, W

        def set_view(self, slot, gray, R, t):
            uploads.append((slot, np.asarray(t, np.float64).copy()))

        def patchmatch(self, refs, srcs, params, seed):
            n = len(refs)
            return (np.zeros((n, self.H, self.W), np.float32), np.zeros((n, self.H, self.W, 3), np.float32),
                    np.zeros((n, self.H, self.W), np.float32))

        def timing(self):
            return {}

        def close(self):
            pass

    monkeypatch.setattr(engine_mod, "Engine", FakeEngine)
    pm = PatchMatchMVS(amvs.Camera(K=scene_a.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7, device=0)
    images = {i: {"gray": scene_a.grays[i], "color": scene_a.colors[i], "shape": scene_a.grays[i].shape}
              for i in range(scene_a.n)}
    poses = scene_a.poses()
    pm._patchmatch_cuda(2, [1, 3], images, poses)
    assert len(uploads) == scene_a.n
    pm._patchmatch_cuda(2, [1, 3], images, poses)
    assert len(uploads) == scene_a.n                               # same dict, same poses: reused
    moved = dict(poses)
    moved[1] = amvs.CameraPose(R=poses[1].R.copy(), t=poses[1].t + np.array([0.01, 0.0, 0.0]))
    pm._patchmatch_cuda(2, [1, 3], images, moved)
    assert len(uploads) == 2 * scene_a.n                           # refined pose: everything re-uploaded
    assert np.allclose(uploads[-scene_a.n + 1][1], moved[1].t)
    pm._patchmatch_cuda(2, [1, 3], dict(images), moved)            # an equal but different dict object
    assert len(uploads) == 3 * scene_a.n


def test_group_launch_plan_is_consistent_across_ranks():
    """PatchMatchMVS._plan_group_launches (the several-rank _sweep_resident): for many scene sizes, world
    sizes, batch caps and source-count patterns every rank issues the same sequence of group gathers, each
    group exactly once and only after the launches that cover its rows; every job of a rank is launched
    exactly once, launches hold consecutive jobs of one source count, stay inside one row group and within
    the cap.  (The RCCL collectives of the ranks must match one for one -- this is the part of the
    multi-GPU path that no one-GPU box can execute with real peers.)"""
    import amvs
    from amvs.parallel import shard
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 4, 5, 7, 8, 16, 17, 32, 33, 64):
        for world in (1, 2, 3, 4, 8):
            for cap in (1, 2, 4, 16):
                counts = rng.choice([2, 3, 4], size=n, p=[0.1, 0.2, 0.7])
                jobs = [(j, list(range(int(counts[j])))) for j in range(n)]
                per = -(-n // world)
                seqs = []
                for rank in range(world):
                    mine = shard(n, rank, world)
                    base = rank * per
                    groups, plan = pm._plan_group_launches(jobs, mine, per, base, cap)
                    assert groups[0][0] == 0 and groups[-1][1] == per
                    assert all(a[1] == b[0] for a, b in zip(groups, groups[1:]))
                    launched, gathered, rows_done = [], [], 0
                    for piece, ready in plan:
                        if piece is not None:
                            assert 1 <= len(piece) <= cap
                            assert piece == list(range(piece[0], piece[0] + len(piece)))
                            assert len({len(jobs[j][1]) for j in piece}) == 1
                            g0 = [g for g, (a, b) in enumerate(groups) if a <= piece[0] - base < b]
                            g1 = [g for g, (a, b) in enumerate(groups) if a <= piece[-1] - base < b]
                            assert g0 == g1 and len(g0) == 1, "a launch straddles a group boundary"
                            launched += piece
                            rows_done = piece[-1] - base + 1
                        for g in ready:
                            assert rows_done >= min(groups[g][1], len(mine)), "group gathered before it was swept"
                            gathered.append(g)
                    assert launched == mine, (n, world, cap, rank)
                    assert gathered == list(range(len(groups))), (n, world, cap, rank, gathered)
                    seqs.append((groups, gathered))
                assert all(s == seqs[0] for s in seqs), "ranks disagree about the collectives"


def test_group_storage_layout_is_contiguous_per_exchange():
    """PatchMatchMVS._storage_row (round 4): the several-rank _sweep_resident lays its maps out [group][rank][row], so
    that (1) every job has its own row inside the world * per rows, (2) the rows of one group form ONE contiguous
    block in which rank r's rows are the r-th equal slice -- the output and (in place) input of one
    all_gather_into_tensor --, and (3) a launch, which never straddles a group, writes consecutive rows."""
    import amvs
    from amvs.parallel import shard
    pm = amvs.PatchMatchMVS.__new__(amvs.PatchMatchMVS)
    for n in (1, 2, 3, 5, 8, 16, 17, 32, 33, 64):
        for world in (1, 2, 3, 4, 8):
            per = -(-n // world)
            jobs = [(j, [0, 1, 2, 3]) for j in range(n)]
            groups_seen = None
            rows = {}
            for rank in range(world):
                mine = shard(n, rank, world)
                groups, plan = pm._plan_group_launches(jobs, mine, per, rank * per, 16)
                assert groups_seen in (None, groups)
                groups_seen = groups
                for piece, _ in plan:
                    if piece is None:
                        continue
                    r = [pm._storage_row(j, per, world, groups) for j in piece]
                    assert r == list(range(r[0], r[0] + len(r))), "a launch must write consecutive rows"
                    rows.update(zip(piece, r))
            # every slot of the padded layout, not only the real jobs
            all_rows = [pm._storage_row(j, per, world, groups_seen) for j in range(world * per)]
            assert sorted(all_rows) == list(range(world * per)), (n, world)
            assert all(rows[j] == all_rows[j] for j in rows)
            for a, b in groups_seen:
                g = b - a
                for rank in range(world):
                    got = [all_rows[rank * per + k] for k in range(a, b)]
                    assert got == list(range(world * a + rank * g, world * a + (rank + 1) * g)), (n, world, a, b, rank)


def test_bench_starts_its_own_ranks_and_reports_a_failing_one(tmp_path):
    """`python bench.py --gpus N` without a launcher starts N fresh child processes itself (bench.launch_ranks)
    before anything touches torch or the GPU, relays rank 0's stdout and fails when a rank fails -- and the
    rank that would then wait in a collective for ever is stopped.  Here (no GPU) the children are a stand-in
    script: the launcher's environment and exit-code handling are what is under test."""
    import subprocess
    import sys
    import textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = tmp_path / "bench.py"
    src = open(os.path.join(root, "bench.py")).read()
    # the real launcher, a stand-in for everything after it
    head = src[:src.index("def main():")]
    fake.write_text(head + textwrap.dedent('''
        def main():
            if "WORLD_SIZE" not in os.environ:
                sys.exit(launch_ranks(int(sys.argv[sys.argv.index("--gpus") + 1])))
            r, w = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
            assert os.environ["LOCAL_RANK"] == str(r) and os.environ["MASTER_ADDR"] == "127.0.0.1"
            assert int(os.environ["MASTER_PORT"]) > 0 and w == 3
            if "--fail" in sys.argv and r == 1:
                sys.exit(7)
            if "--fail" in sys.argv:
                time.sleep(60)                   # a rank stuck in a collective
            print(json.dumps({"rank": r, "world": w}), flush=True)

        main()
    '''))
    ok = subprocess.run([sys.executable, str(fake), "--gpus", "3"], capture_output=True, text=True, timeout=60)
    assert ok.returncode == 0, ok.stderr
    assert ok.stdout.strip() == '{"rank": 0, "world": 3}'          # rank 0's line only; the others went to stderr
    assert '{"rank": 2, "world": 3}' in ok.stderr
    import time
    t0 = time.time()
    bad = subprocess.run([sys.executable, str(fake), "--gpus", "3", "--fail"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 7 and time.time() - t0 < 30, (bad.returncode, bad.stderr)
    assert "rank 1 exited with 7" in bad.stderr


@pytest.mark.gpu
def test_bench_two_ranks_self_launched_on_one_gpu():
    """The driver's command shape, `python bench.py --gpus 2 ...` with no launcher and no WORLD_SIZE, on a
    one-GPU box: both ranks on cuda:0, the exchange over gloo (AMVS_BENCH_BACKEND / AMVS_BENCH_ONE_DEVICE
    are rehearsal switches; the real run is one rank per GPU over RCCL).  One valid JSON line, the strong-
    scaling scene, two batches per step with their contiguous all_gather_into_tensor blocks, and the
    bench's own closing asserts (own rows bit for bit, peer rows populated)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, AMVS_BENCH_BACKEND="gloo", AMVS_BENCH_ONE_DEVICE="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--scene-views", "16", "--height", "270", "--width", "480", "--no-cpu-baseline"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["scaling"] == "strong"
    assert rec["config"]["views_per_gpu"] == 8 and rec["config"]["batches_per_step"] == 2
    assert rec["value"] > 0 and rec["dense_points"]["raw"] > 0
