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


def _oracle_backend(scene, patch, iters, samples, seed):
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
                                     [scene.t[i] for i in srcs], patch)
            a, b, e = ctx.patchmatch(iters, samples, scene.depth_min, scene.depth_max, seed, ref)
            d.append(a); n.append(b); c.append(e)
        return np.stack(d), np.stack(n), np.stack(c)
    return run


def _reconstruct(scene, world_tag, use_engine=False):
    """PatchMatchMVS.reconstruct; without a GPU the device sweep is replaced by the oracle backend."""
    import amvs
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    pm = PatchMatchMVS(amvs.Camera(K=scene.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                       num_iterations=1, num_samples=2, min_views=2, seed=5, views_per_batch=2, device=0)
    if not use_engine:
        backend = _oracle_backend(scene, 7, 1, 2, 5)
        pm._ensure_engine = lambda images, poses, indices: setattr(pm, "_slot", {i: i for i in indices})
        pm._run_batch = lambda eng, batch: backend(batch)
        pm.device_fusion = False        # host maps + NumPy fusion (bit-identical to the device path)
    # keep the synthetic scene's depth range (the reference would estimate it from sparse points)
    pm._estimate_depth_range = lambda poses, sparse: None
    pm.depth_min, pm.depth_max = scene.depth_min, scene.depth_max
    return pm.reconstruct([{"image": c} for c in scene.colors], scene.poses())


def _worker(rank, world, port, q, use_engine=False):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      AMVS_ORACLE_THREADS="2")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from conftest import GoldenScene
    from amvs import parallel
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # 1. raw collective: uneven shards (5 items over 2 ranks -> 3 + 2)
        mine = parallel.shard(5, rank, world)
        local = torch.tensor([[float(j), j * 10.0, j + 0.5] for j in mine], dtype=torch.float32).reshape(len(mine), 3)
        full = parallel.allgather_packed(local, 5, 3)
        want = torch.tensor([[float(j), j * 10.0, j + 0.5] for j in range(5)], dtype=torch.float32)
        assert torch.equal(full, want), (rank, full)
        # 2. the sharded reconstruct path
        pts, cols = _reconstruct(GoldenScene("scene_a"), f"rank{rank}", use_engine)
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


def _run_two_ranks(use_engine):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_engine)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


@pytest.mark.gpu
@pytest.mark.timeout(300)
def test_reconstruct_on_gpu_single_and_two_ranks(scene_a):
    """The drop-in class end to end on the HIP engine: (1) its fused cloud equals the one obtained
    with the CPU oracle as sweep backend (the sweep is bit-exact, the host geometry is shared);
    (2) two ranks (gloo, both on cuda:0) sharding the views return the same cloud on every rank."""
    gpu_pts, gpu_cols = _reconstruct(scene_a, "gpu", use_engine=True)
    cpu_pts, cpu_cols = _reconstruct(scene_a, "oracle", use_engine=False)
    assert gpu_pts.shape[0] > 0
    assert np.array_equal(gpu_pts, cpu_pts) and np.array_equal(gpu_cols, cpu_cols)
    for rank, pts, cols in _run_two_ranks(True):
        assert np.array_equal(pts, gpu_pts), f"rank {rank}"
        assert np.array_equal(cols, gpu_cols)
