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


def sharded_hierarchical_merge(scene, total_frames, overlap_thresh_factor=0.025, group=None):
    """hierarchical_merge (graph_utils.py:989-1012) of ONE episode whose frames are spread over the ranks:
    rank r holds the frame window [r * chunk, (r + 1) * chunk) with chunk a power of two (scene.set_frame_window), has
    fused its own frames, and calls this.  Rank-local tree levels first, then log2(ranks) cross-rank levels in which
    the odd list of every pair travels to the rank holding the even one (point clouds, over torch.distributed send /
    recv: RCCL on the GPUs, gloo in the CPU test); rank 0 ends with the instances of the whole episode, bit-identical
    to a single-process hmsg_merge_instances.  Returns True on the rank that holds the result."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    th, lists, idx = scene.merge_tree_local(total_frames)
    stride = 1                                   # ranks between the owners of adjacent lists at this level
    active = True
    if lists == 1:                               # one rank held every frame
        scene.merge_tree_join([], th, final_pass=True)
        return True
    while lists > 1:
        nxt = (lists + 1) // 2
        if active:
            if idx % 2 == 0:
                if idx + 1 < lists:
                    src = rank + stride
                    meta = torch.zeros(1, dtype=torch.int64)
                    dist.recv(meta, src=src, group=group)
                    sizes = torch.zeros(int(meta.item()), dtype=torch.int64)
                    if len(sizes):
                        dist.recv(sizes, src=src, group=group)
                    pts = torch.zeros((int(sizes.sum().item()), 3), dtype=torch.float64)
                    if len(pts):
                        dist.recv(pts, src=src, group=group)
                    off = np.concatenate([[0], np.cumsum(sizes.numpy())])
                    clouds = [pts.numpy()[off[k]:off[k + 1]] for k in range(len(sizes))]
                    scene.merge_tree_join(clouds, th, final_pass=(nxt == 1))
                # (an even list without a partner is carried to the next level unchanged)
            else:
                dst = rank - stride
                clouds = scene.instances()
                sizes = torch.tensor([len(c) for c in clouds], dtype=torch.int64)
                dist.send(torch.tensor([len(clouds)], dtype=torch.int64), dst=dst, group=group)
                if len(clouds):
                    dist.send(sizes, dst=dst, group=group)
                pts = torch.from_numpy(np.ascontiguousarray(np.concatenate(clouds) if len(clouds) else np.zeros((0, 3))))
                if len(pts):
                    dist.send(pts, dst=dst, group=group)
                active = False
        idx //= 2
        stride *= 2
        lists = nxt
        if lists > 1:
            th -= overlap_thresh_factor * (lists - 2) / max(1, lists - 1)
    return active


def allreduce_feature_sums(scene, group=None, device=None):
    """One episode fused in disjoint frame windows: sum the per-voxel feature sums and frame counters of
    all ranks (all-reduce of V * (D + 1) * 4 bytes over RCCL / gloo) and install the result on every rank.  Counters
    are exact; the float32 sums agree with a single-process build up to summation order (<= 1e-5)."""
    import torch
    import torch.distributed as dist
    sums, cnt = scene.feature_sums()
    dev = device if device is not None else torch.device("cpu")
    ts = torch.from_numpy(sums).to(dev)
    tc = torch.from_numpy(cnt.astype(np.int64)).to(dev)
    dist.all_reduce(ts, group=group)
    dist.all_reduce(tc, group=group)
    scene.set_feature_sums(ts.cpu().numpy(), tc.cpu().numpy().astype(np.uint32))
