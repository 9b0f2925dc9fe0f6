"""Multi-GPU glue: scenes are independent, so the build has no data-path collective; for
cross-scene retrieval the per-rank node tables are all-gathered (counts first, then a payload padded to the
largest table) over `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests) and every rank answers its share of the queries on the global table.  Global node index = prefix
offset of the owning rank + local index; room ids are made global the same way."""
from __future__ import annotations

import numpy as np


def gather_node_tables(feats: np.ndarray, rooms: np.ndarray, n_rooms_local: int, device=None):
    """feats f64 [n, D], rooms i32 [n] (local room ids) -> (global feats [N, D], global room ids [N],
    node offsets per rank [world+1], room offsets per rank [world+1])."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = device if device is not None else torch.device("cpu")
    D = feats.shape[1]
    meta = torch.tensor([feats.shape[0], n_rooms_local], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    counts = [int(m[0].item()) for m in metas]
    nrooms = [int(m[1].item()) for m in metas]
    node_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    room_off = np.concatenate([[0], np.cumsum(nrooms)]).astype(np.int64)
    nmax = max(max(counts), 1)
    pay = torch.zeros((nmax, D + 1), dtype=torch.float64, device=dev)
    if feats.shape[0]:
        pay[: feats.shape[0], :D] = torch.from_numpy(np.ascontiguousarray(feats, np.float64)).to(dev)
        pay[: feats.shape[0], D] = torch.from_numpy((rooms.astype(np.int64) + room_off[rank]).astype(np.float64)).to(dev)
    allp = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(allp, pay)
    tab = torch.cat([allp[r][: counts[r]] for r in range(world)]).cpu().numpy()
    return np.ascontiguousarray(tab[:, :D]), tab[:, D].astype(np.int32), node_off, room_off


def shard_queries(n_queries: int, rank: int, world: int):
    """Round-robin share of the query batch for this rank."""
    return list(range(rank, n_queries, world))
