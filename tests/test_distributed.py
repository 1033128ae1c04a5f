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
    from oracle import oracle

    def run(jobs):
        d, n, c = [], [], []
        for ref, srcs in jobs:
            ctx = scene.oracle_ctx(ref, srcs, patch)
            a, b, e = ctx.patchmatch(iters, samples, scene.depth_min, scene.depth_max, seed, ref)
            d.append(a); n.append(b); c.append(e)
        return np.stack(d), np.stack(n), np.stack(c)
    return run


def _reconstruct(scene, world_tag):
    """PatchMatchMVS.reconstruct with the device sweep replaced by the oracle backend."""
    import amvs
    from amvs.core.mvs_patchmatch import PatchMatchMVS
    pm = PatchMatchMVS(amvs.Camera(K=scene.K.copy(), dist=np.zeros(5)), scale=1.0, patch_size=7,
                       num_iterations=1, num_samples=2, min_views=2, seed=5, views_per_batch=2)
    backend = _oracle_backend(scene, 7, 1, 2, 5)
    pm._ensure_engine = lambda images, poses, indices: setattr(pm, "_slot", {i: i for i in indices})
    pm._run_batch = lambda eng, batch: backend(batch)
    # keep the synthetic scene's depth range (the reference would estimate it from sparse points)
    pm._estimate_depth_range = lambda poses, sparse: None
    pm.depth_min, pm.depth_max = scene.depth_min, scene.depth_max
    return pm.reconstruct([{"image": c} for c in scene.colors], scene.poses())


def _worker(rank, world, port, q):
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
        pts, cols = _reconstruct(GoldenScene("scene_a"), f"rank{rank}")
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
    import torch.multiprocessing as mp
    single_pts, single_cols = _reconstruct(scene_a, "single")
    assert single_pts.shape[0] > 0 and single_pts.shape[1] == 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, pts, cols in results:
        assert np.array_equal(pts, single_pts), f"rank {rank} cloud differs from the single-process cloud"
        assert np.array_equal(cols, single_cols)
