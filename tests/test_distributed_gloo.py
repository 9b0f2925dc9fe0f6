"""N>1 path on CPU: 2 ranks (gloo) all-gather their node tables and answer their share of the queries on the
global table; the union of the answers must equal the single-table answer bit for bit."""
import os
import socket

import numpy as np
import pytest

from tests import parity_common as PC

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


def _tables(rank, D=24):
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    n = 40 + 13 * rank
    return rng.standard_normal((n, D)) * 0.1, rng.integers(0, 3, size=n).astype(np.int32), 3


def _queries(D=24, Q=10):
    rng = np.random.Generator(np.random.PCG64(7))
    T = rng.standard_normal((Q, 2, D)).astype(np.float32)
    lists = [sorted(rng.choice(6, size=2, replace=False).tolist()) for _ in range(Q)]
    return T, lists


def _worker(rank, world, port, out):
    import torch.distributed as dist
    from holoagent_amd._lib import HmsgLib, NodeIndex
    from holoagent_amd.dist import gather_node_tables, shard_queries
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    feats, rooms, nr = _tables(rank)
    g_feats, g_rooms, node_off, room_off = gather_node_tables(feats, rooms, nr)
    T, lists = _queries()
    mine = shard_queries(len(lists), rank, world)
    ix = NodeIndex(g_feats, g_rooms, lib_=HmsgLib(PC.EMU_PATH))
    idx, room, score = ix.query_objects(T[mine], np.zeros(len(mine), np.int32), [lists[q] for q in mine], 4)
    ix.close()
    np.savez(out % rank, idx=idx, score=score, mine=np.array(mine), node_off=node_off, n=g_feats.shape[0])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_allgather_retrieval(tmp_path):
    import torch.multiprocessing as mp
    from holoagent_amd._lib import HmsgLib, NodeIndex
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "r%d.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    # single-table answer
    tabs = [_tables(r) for r in range(2)]
    feats = np.concatenate([t[0] for t in tabs])
    rooms = np.concatenate([tabs[0][1], tabs[1][1] + 3]).astype(np.int32)
    T, lists = _queries()
    ix = NodeIndex(feats, rooms, lib_=HmsgLib(PC.EMU_PATH))
    idx, room, score = ix.query_objects(T, np.zeros(len(lists), np.int32), lists, 4)
    ix.close()
    seen = set()
    for r in range(2):
        z = np.load(out % r)
        assert int(z["n"]) == feats.shape[0]
        assert list(z["node_off"]) == [0, tabs[0][0].shape[0], feats.shape[0]]
        for j, q in enumerate(z["mine"]):
            np.testing.assert_array_equal(z["idx"][j], idx[q])
            np.testing.assert_array_equal(z["score"][j], score[q])
            seen.add(int(q))
    assert seen == set(range(len(lists)))


def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without a launcher re-executes itself under torch.distributed.run with one process
    per GPU (checked here without GPUs: the ranks rendezvous over gloo and report the world size)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, HMSG_BENCH_SPAWN_ONLY="1", MASTER_PORT="29613")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["rank_sum"] == 3


# ---- config 5 building block: the hierarchical merge tree of ONE episode sharded over the ranks ---------------------
def _episode(n_frames=4, size=(64, 48)):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=11, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=size[0], height=size[1],
                     n_frames=n_frames, n_masks=8, feat_dim=16, yaw_step_deg=25.0 if n_frames <= 16 else 11.0)
    scn = SynthScene(spec)
    return [scn.frame(i) for i in range(n_frames)]


def _build(L, frames, window=None):
    """Scene with the whole map; features / masks of frames[window] only (all frames when None)."""
    S = PC.stack_frames(frames)
    big = frames[0]["depth"].shape[1] >= 320          # (the 320 x 240 episodes are dense enough for a real outlier filter)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, merge_type=1, outlier_nb_points=200 if big else 20, outlier_radius=0.5 if big else 0.3))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    a, b = window if window is not None else (0, len(frames))
    if window is not None:
        sc.set_frame_window(a)
    sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
    sc.fuse_frames()
    return sc


def _merge_worker(rank, world, port, out, n_frames, chunk, size=(64, 48)):
    import torch.distributed as dist
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.dist import sharded_hierarchical_merge
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _episode(n_frames, size)
    sc = _build(HmsgLib(PC.EMU_PATH), frames, (rank * chunk, min(n_frames, (rank + 1) * chunk)))
    from holoagent_amd.dist import allreduce_feature_sums
    allreduce_feature_sums(sc)                       # every rank now holds the whole episode's voxel features
    holds = sharded_hierarchical_merge(sc, len(frames))
    assert holds == (rank == 0)
    if rank == 0:
        inst = sc.instances()
        sc.pool_instances()
        feats, counter = sc.map_feats(counter=True)
        np.savez(out, sizes=np.array([len(c) for c in inst]), pts=np.concatenate(inst) if inst else np.zeros((0, 3)),
                 map_feats=feats, counter=counter, pooled=sc.instance_feats())
    dist.barrier()
    sc.close()
    dist.destroy_process_group()


# (world, frames, frames per rank): two even ranks; THREE ranks with a shorter last window (the last rank stops its local
# tree two levels below the others and is carried up); four ranks on request
_slow = pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the simulator (HMSG_EMU_SLOW=1)")


@pytest.mark.parametrize("world,n_frames,chunk,size", [
    (2, 4, 2, (64, 48)), (3, 5, 2, (64, 48)),
    pytest.param(4, 4, 1, (64, 48), marks=_slow),
    pytest.param(3, 10, 4, (64, 48), marks=_slow),
    # a 32-frame 320 x 240 episode over two and over four ranks (last run: profiles/r04_gloo_episode_320x240.txt)
    pytest.param(2, 32, 16, (320, 240), marks=_slow),
    pytest.param(4, 32, 8, (320, 240), marks=_slow)])
def test_sharded_hierarchical_merge_equals_single_process(tmp_path, world, n_frames, chunk, size):
    """Frames of one episode split over the ranks: all-reduce of the voxel feature sums, rank-local merge-tree levels +
    cross-rank joins (torch.distributed send / recv), pooling on the root == one process over all frames (instances bit
    for bit, features within 1e-5)."""
    import torch.multiprocessing as mp
    from holoagent_amd._lib import HmsgLib
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "merged.npz")
    mp.spawn(_merge_worker, args=(world, port, out, n_frames, chunk, size), nprocs=world, join=True)
    sc = _build(HmsgLib(PC.EMU_PATH), _episode(n_frames, size))
    sc.merge_instances()
    ref = sc.instances()
    sc.pool_instances()
    ref_feats, ref_counter = sc.map_feats(counter=True)
    ref_pooled = sc.instance_feats()
    sc.close()
    z = np.load(out)
    assert len(ref) > 3 and z["sizes"].tolist() == [len(c) for c in ref]
    assert np.array_equal(z["pts"], np.concatenate(ref))
    # voxel features all-reduced over the ranks: counters exact, float32 sums up to summation order; pooled features
    # of the episode within the north-star tolerance
    assert np.array_equal(z["counter"], ref_counter)
    np.testing.assert_allclose(z["map_feats"], ref_feats, rtol=0, atol=1e-5)
    np.testing.assert_allclose(z["pooled"], ref_pooled, rtol=0, atol=1e-5)


def test_tree_schedule_pairs_every_rank_layout():
    """The send / recv schedule of sharded_hierarchical_merge, simulated without processes for every (ranks, frames per
    rank, total) with power-of-two chunks and a shorter last window: each level's sends match its receives, exactly one
    rank ends up with the result, and the number of joins equals ranks - 1."""
    def local(total, first, n):
        lists, off = total, first
        while lists > 1:
            if n == 1 or (off & 1) or ((n & 1) and off + n < lists):
                break
            n = (n + 1) // 2
            off >>= 1
            lists = (lists + 1) // 2
        return lists, off
    for chunk in (1, 2, 4, 8):
        for world in range(1, 9):
            for total in range((world - 1) * chunk + 1, world * chunk + 1):
                state = [local(total, r * chunk, min(total, (r + 1) * chunk) - r * chunk) for r in range(world)]
                target = min(l for l, _ in state)
                owner = {}
                for r, (l, i) in enumerate(state):
                    while l > target:
                        assert i == l - 1 and l % 2 == 1, (chunk, world, total, r)
                        i //= 2
                        l = (l + 1) // 2
                    assert i not in owner
                    owner[i] = r
                lists = target
                assert sorted(owner) == list(range(lists)), (chunk, world, total)
                joins = 0
                while lists > 1:
                    sends = {owner[k - 1]: owner[k] for k in owner if k % 2 == 1}
                    recvs = {owner[k]: owner[k + 1] for k in owner if k % 2 == 0 and k + 1 < lists}
                    assert sends == recvs
                    joins += len(sends)
                    owner = {k // 2: r for k, r in owner.items() if k % 2 == 0}
                    lists = (lists + 1) // 2
                assert joins == world - 1 and owner == {0: 0}
