"""Multi-GPU glue: scenes are independent, so the build has no data-path collective; for
cross-scene retrieval the per-rank node tables are all-gathered (counts first, then a payload padded to the
largest table) over `torch.distributed` (backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU
tests) and every rank answers its share of the queries on the global table.  Global node index = prefix
offset of the owning rank + local index; room ids are made global the same way."""
from __future__ import annotations

import numpy as np


def gather_node_tables(feats, rooms: np.ndarray, n_rooms_local: int, device=None):
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
        # (feats: numpy, or a torch tensor already on the device -- no detour over the host then)
        pay[: feats.shape[0], :D] = (feats.to(dev, torch.float64) if hasattr(feats, "data_ptr")
                                     else torch.from_numpy(np.ascontiguousarray(feats, np.float64)).to(dev))
        pay[: feats.shape[0], D] = torch.from_numpy((rooms.astype(np.int64) + room_off[rank]).astype(np.float64)).to(dev)
    allp = [torch.empty_like(pay) for _ in range(world)]
    dist.all_gather(allp, pay)
    tab = torch.cat([allp[r][: counts[r]] for r in range(world)]).cpu().numpy()
    return np.ascontiguousarray(tab[:, :D]), tab[:, D].astype(np.int32), node_off, room_off


def gather_node_tables_device(scene, n_rooms_local: int, comm=None, device_id: int = 0, group=None):
    """The same exchange step behind the C ABI (include/hmsg.h: hmsg_allgather_nodes): the ranks' node tables go from HBM to
    HBM through librccl, called by the library itself on the handle's stream, and come back as ONE resident retrieval index --
    no torch tensor, no host copy of the payload.  `comm`: a holoagent_amd._lib.Comm (made once per process; by default from
    the initialised torch.distributed job: rank 0's 128-byte id travels through the process group).
    Returns (NodeIndex over the global table, node_off [world + 1], room_off [world + 1], comm)."""
    from ._lib import Comm
    if comm is None:
        import torch.distributed as dist
        comm = Comm.from_torch(device_id, scene.L, group) if dist.is_initialized() else Comm.single(device_id, scene.L)
    ix, node_off, room_off = scene.allgather_nodes(comm, n_rooms_local)
    return ix, node_off, room_off, comm


def shard_queries(n_queries: int, rank: int, world: int):
    """Round-robin share of the query batch for this rank."""
    return list(range(rank, n_queries, world))


def _next_threshold(th, factor, lists):
    """threshold after a level that left `lists` lists (graph_utils.py:1001-1003)"""
    return th - factor * (lists - 2) / max(1, lists - 1) if lists > 1 else th


def sharded_hierarchical_merge(scene, total_frames, overlap_thresh_factor=0.025, group=None, device=None):
    """hierarchical_merge (graph_utils.py:989-1012) of ONE episode whose frames are spread over the ranks:
    rank r holds a frame window that is a subtree of the merge tree (first frame a multiple of a power of two >= the
    window length: scene.set_frame_window; equal power-of-two chunks with a shorter last one qualify), has fused its own
    frames, and calls this.

    1. rank-local tree levels (hmsg_merge_tree_local): every rank ends with ONE list and reports the level it stopped at;
    2. the ranks agree on the level (all-gather of (lists, index)): a rank that stopped lower holds the last, unpaired
       list of its level and carries it up unchanged, as merge_adjacent_frames does with an odd last list;
    3. cross-rank levels: the owner of list 2k+1 sends its clouds to the owner of list 2k, which merges [mine ++ theirs]
       (hmsg_merge_tree_join); an unpaired last list stays where it is.  Owners are tracked per list, so any number of
       ranks works (3, 5, 6, ... not only powers of two).

    `device`: where the exchanged tensors live -- None / cpu with the gloo backend (CPU tests), the rank's GPU with the
    nccl backend (RCCL: send / recv need device tensors; the clouds then go from HBM to HBM, hmsg_merge_tree_join reads
    the receive buffer in place).  The root (rank 0) ends with the instances of the whole episode, bit-identical to a
    single-process hmsg_merge_instances.  Returns True on the rank that holds the result."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = device if device is not None else torch.device("cpu")
    th, lists, idx = scene.merge_tree_local(total_frames)
    meta = torch.tensor([lists, idx], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    state = [(int(m[0].item()), int(m[1].item())) for m in metas]
    target = min(l for l, _ in state)
    owner = {}                                   # list index at the common level -> rank
    for r, (l, i) in enumerate(state):
        t = th if r == rank else None
        while l > target:
            if not (i == l - 1 and l % 2 == 1):
                raise ValueError("sharded_hierarchical_merge: rank %d's frame window is not a subtree of the merge tree "
                                 "(list %d of %d cannot be carried up)" % (r, i, l))
            i //= 2
            l = (l + 1) // 2
            if t is not None:
                t = _next_threshold(t, overlap_thresh_factor, l)
        if i in owner:
            raise ValueError("sharded_hierarchical_merge: ranks %d and %d both hold list %d" % (owner[i], r, i))
        owner[i] = r
        if r == rank:
            th, lists, idx = t, l, i
    if sorted(owner) != list(range(lists)):
        raise ValueError("sharded_hierarchical_merge: the ranks hold lists %s of %d" % (sorted(owner), lists))
    if lists == 1:                               # one rank held every frame
        if owner[0] == rank:
            scene.merge_tree_join([], th, final_pass=True)
        return owner[0] == rank
    active = True
    while lists > 1:
        nxt = (lists + 1) // 2
        if active and idx % 2 == 1:              # my list is the odd one of its pair: it travels
            dst = owner[idx - 1]
            sizes = scene.instance_sizes()
            dist.send(torch.tensor([len(sizes)], dtype=torch.int64, device=dev), dst=dst, group=group)
            if len(sizes):
                dist.send(torch.from_numpy(sizes).to(dev), dst=dst, group=group)
                total = int(sizes.sum())
                if total:
                    pts = torch.empty((total, 3), dtype=torch.float64, device=dev)
                    scene.instance_points_into(pts)          # HBM -> HBM when dev is the GPU
                    dist.send(pts, dst=dst, group=group)
            active = False
        elif active and idx + 1 < lists:         # I hold the even one: merge [mine ++ theirs]
            src = owner[idx + 1]
            n = torch.zeros(1, dtype=torch.int64, device=dev)
            dist.recv(n, src=src, group=group)
            sizes = torch.zeros(int(n.item()), dtype=torch.int64, device=dev)
            pts = None
            if len(sizes):
                dist.recv(sizes, src=src, group=group)
                total = int(sizes.sum().item())
                if total:
                    pts = torch.empty((total, 3), dtype=torch.float64, device=dev)
                    dist.recv(pts, src=src, group=group)
            scene.merge_tree_join_raw(sizes.cpu().numpy(), pts, th, final_pass=(nxt == 1))
        # (an even list without a partner is carried to the next level unchanged)
        owner = {k // 2: r for k, r in owner.items() if k % 2 == 0}
        idx //= 2
        lists = nxt
        th = _next_threshold(th, overlap_thresh_factor, lists)
    return active


def allreduce_feature_sums(scene, group=None, device=None):
    """One episode fused in disjoint frame windows: sum the per-voxel feature sums and frame counters of
    all ranks (all-reduce of V * (D + 1) * 4 bytes over RCCL / gloo) and install the result on every rank.  Counters
    are exact; the float32 sums agree with a single-process build up to summation order (<= 1e-5).
    With `device` = the rank's GPU the buffers never leave HBM: the library copies its sums into the collective's
    tensors and reads the reduced ones back by device pointer."""
    import torch
    import torch.distributed as dist
    dev = device if device is not None else torch.device("cpu")
    V, D = scene.map_size(), scene.cfg.feat_dim
    ts = torch.empty((V, D), dtype=torch.float32, device=dev)
    tc32 = torch.empty((V,), dtype=torch.int32, device=dev)      # (the library's counters are u32; int32 views them bit for bit)
    scene.feature_sums_into(ts, tc32)
    tc = tc32.to(torch.int64)
    dist.all_reduce(ts, group=group)
    dist.all_reduce(tc, group=group)
    # (under the nccl backend all_reduce only makes torch's current stream wait: the library works on its own stream, so the
    #  tensors -- and the int32 conversion below, a kernel -- must be complete before their pointers are handed over;
    #  _lib._ptr drains torch's stream for every device tensor, this is the explicit form of the same guarantee)
    tc32 = tc.to(torch.int32).contiguous()
    if dev.type == "cuda":
        torch.cuda.current_stream(dev).synchronize()
    scene.set_feature_sums_from(ts, tc32)


def allreduce_feature_sums_device(scene, comm):
    """The same all-reduce behind the C ABI (include/hmsg.h: hmsg_allreduce_feature_sums): librccl on the handle's own
    buffers and stream, in place."""
    scene.allreduce_feature_sums(comm)
