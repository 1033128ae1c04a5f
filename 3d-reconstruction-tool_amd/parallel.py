"""Multi-GPU plumbing: one process per GPU, reference views sharded over ranks.

The reference has no distributed code; its outer loop over reference
views (mvs_patchmatch.py:104-123) has independent iterations, so the views are split
into contiguous blocks, every rank keeps all images resident (source sets cross shard
boundaries) and the only exchange is one all-gather of the per-view result maps
(depth 4 B + normal 12 B + confidence 4 B per pixel) before fusion.  torch.distributed
is the transport: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU (tests).
"""
import os

import numpy as np


def local_device() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


def _dist():
    try:
        import torch.distributed as dist
    except Exception:  # noqa: BLE001
        return None
    return dist if dist.is_available() and dist.is_initialized() else None


def rank_world(group=None):
    dist = _dist()
    if dist is None:
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def shard(n_items: int, rank: int, world: int):
    """Contiguous block of ceil(n/world) items per rank."""
    per = (n_items + world - 1) // world
    return list(range(min(rank * per, n_items), min((rank + 1) * per, n_items)))


def shard_sizes(n_items: int, world: int):
    return [len(shard(n_items, r, world)) for r in range(world)]


def allgather_packed(local, n_items: int, width: int, group=None, device=None):
    """All-gather rows of `width` float32: rank r contributes the rows of shard(n_items, r).

    local: torch tensor (len(shard), width) on `device` (CUDA tensor -> RCCL, CPU -> gloo).
    Returns a tensor (n_items, width) identical on every rank.  Shards are padded to the
    common size ceil(n/world) so a single all_gather_into_tensor moves everything.
    """
    import torch
    import torch.distributed as dist

    rank, world = rank_world(group)
    per = (n_items + world - 1) // world
    dev = local.device if device is None else device
    send = torch.zeros((per, width), dtype=torch.float32, device=dev)
    send[: local.shape[0]] = local
    recv = torch.empty((world * per, width), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    sizes = shard_sizes(n_items, world)
    parts = [recv[r * per: r * per + sizes[r]] for r in range(world)]
    return torch.cat(parts, dim=0)


def allgather_maps(local_maps: dict, n_items: int, shape, group, map_type, device_id=None):
    """Host-side convenience used by PatchMatchMVS._sweep: {item index -> map_type} of this
    rank's shard in, all items out.  Packs depth|normal|confidence = 5 floats per pixel.
    device_id: the GPU the caller's engine runs on (RCCL gathers device tensors; defaults to
    LOCAL_RANK only when the caller does not say)."""
    import torch

    H, W = shape
    rank, world = rank_world(group)
    mine = shard(n_items, rank, world)
    use_cuda = torch.distributed.get_backend(group) == "nccl"
    dev = torch.device("cuda", local_device() if device_id is None else int(device_id)) if use_cuda \
        else torch.device("cpu")
    rows = np.zeros((len(mine), 5 * H * W), np.float32)
    for n, j in enumerate(mine):
        m = local_maps[j]
        rows[n, : H * W] = m.depth.ravel()
        rows[n, H * W: 4 * H * W] = m.normal.ravel()
        rows[n, 4 * H * W:] = m.confidence.ravel()
    full = allgather_packed(torch.from_numpy(rows).to(dev), n_items, 5 * H * W, group).cpu().numpy()
    out = {}
    for j in range(n_items):
        out[j] = map_type(depth=full[j, : H * W].reshape(H, W).copy(),
                          normal=full[j, H * W: 4 * H * W].reshape(H, W, 3).copy(),
                          confidence=full[j, 4 * H * W:].reshape(H, W).copy())
    return out
