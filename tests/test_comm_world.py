"""The C-ABI exchange steps with MORE THAN ONE RANK (include/hmsg.h: hmsg_comm_create, hmsg_allgather_nodes,
hmsg_allreduce_feature_sums; holoagent_amd/csrc/hmsg_comm.hip).  No multi-GPU box is ours to drive and the kernel simulator has
no RCCL, so the six nccl* symbols the library resolves come from a TEST DOUBLE here (tests/rccl_double: shared memory between the
processes of this machine, HMSG_RCCL_LIB) -- what runs is the library's own code around them: counts, padding to the largest
table, the per-rank un-padding, room_off shifting, a rank without nodes, the agreement before the first collective.

  * 2 and 3 ranks with uneven node tables (one rank with none): every rank's gathered index answers like ONE index over the
    concatenated tables (offsets, rooms shifted by room_off), bit for bit;
  * a rank whose table is invalid makes EVERY rank fail (nobody is left waiting in a collective);
  * hmsg_allreduce_feature_sums over 2 ranks' frame windows against the one-process fusion of all frames."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import parity_common as PC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE_SRC = os.path.join(ROOT, "tests", "rccl_double", "rccl_double.cpp")
DOUBLE = os.path.join(ROOT, "tests", "rccl_double", "librccl_double.so")

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


def _double():
    if not os.path.exists(DOUBLE) or os.path.getmtime(DOUBLE) < os.path.getmtime(DOUBLE_SRC):
        subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-o", DOUBLE, DOUBLE_SRC, "-lpthread", "-lrt"], check=True)
    return DOUBLE


def _scene_frames(seed, n_frames=4):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=seed, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                     n_frames=n_frames, n_masks=8, feat_dim=16, yaw_step_deg=25.0)
    scn = SynthScene(spec)
    return [scn.frame(i) for i in range(n_frames)]


def _queries(D=16, Q=9, n_rooms_total=5):
    rng = np.random.Generator(np.random.PCG64(7))
    T = rng.standard_normal((Q, 2, D)).astype(np.float32)
    lists = [sorted(rng.choice(n_rooms_total, size=2, replace=False).tolist()) for _ in range(Q)]
    return T, lists


def _wait_id(path):
    import time
    for _ in range(20000):
        if os.path.exists(path):
            return open(path, "rb").read()
        time.sleep(0.005)
    raise RuntimeError("no communicator id")


def _gather_worker(rank, world, tmp, empty_rank, bad_rank):
    """one rank: a small scene of its own, object nodes in `1 + rank` rooms, the exchange, its answers to every query"""
    os.environ["HMSG_RCCL_LIB"] = DOUBLE
    from holoagent_amd._lib import Comm, HmsgError, HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    idp = os.path.join(tmp, "id.bin")
    if rank == 0:
        uid = Comm.unique_id(lib_=L)
        open(idp + ".tmp", "wb").write(uid)
        os.replace(idp + ".tmp", idp)
    comm = Comm.create(_wait_id(idp), rank, world, lib_=L)
    frames = _scene_frames(20 + rank)
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=20, outlier_radius=0.3, feat_dbscan_min=8))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    n_rooms = 1 + rank
    fl = sc.segment_floors()
    # rooms: vertical strips of the scene's footprint (every object lands in one of them)
    xs = np.linspace(-10.0, 10.0, n_rooms + 1)
    regions = [np.array([[x, z] for x in np.arange(xs[i], xs[i + 1], 0.1) for z in np.arange(-10.0, 10.0, 0.1)]) for i in range(n_rooms)]
    if rank == empty_rank:
        nodes = sc.build_object_nodes([], [], [0] * n_rooms, regions, None)              # no storey: no node
        assert len(nodes) == 0
    else:
        nodes = sc.build_object_nodes([f["zero_level"] for f in fl], [f["height"] for f in fl], [0] * n_rooms, regions, None)
        assert len(nodes) >= 2
    nodes, emb = sc.nodes(embeddings=True)
    np.savez(os.path.join(tmp, "table%d.npz" % rank), emb=emb, room=np.array([int(n["room"]) for n in nodes], np.int32), n_rooms=n_rooms)
    if rank == bad_rank:
        try:
            sc.allgather_nodes(comm, 0)                                                  # its nodes' rooms lie outside a table of 0 rooms
            ok = "no error"
        except HmsgError as e:
            ok = "failed: " + str(e)
        open(os.path.join(tmp, "res%d.txt" % rank), "w").write(ok)
        comm.close()
        sc.close()
        return
    try:
        ix, node_off, room_off = sc.allgather_nodes(comm, n_rooms)
    except HmsgError as e:
        open(os.path.join(tmp, "res%d.txt" % rank), "w").write("failed: " + str(e))
        comm.close()
        sc.close()
        return
    T, lists = _queries(n_rooms_total=int(room_off[-1]))
    idx, room, score = ix.query_objects(T, np.zeros(len(lists), np.int32), lists, 4)
    np.savez(os.path.join(tmp, "ans%d.npz" % rank), idx=idx, room=room, score=score, node_off=node_off, room_off=room_off, n=ix.N)
    ix.close()
    comm.close()
    sc.close()


def _spawn(fn, world, *args):
    import torch.multiprocessing as mp
    mp.spawn(fn, args=(world,) + args, nprocs=world, join=True)


@pytest.mark.parametrize("world,empty_rank", [(2, -1), (3, 1)])
def test_allgather_nodes_with_several_ranks(tmp_path, world, empty_rank):
    from holoagent_amd._lib import HmsgLib, NodeIndex
    _double()
    _spawn(_gather_worker, world, str(tmp_path), empty_rank, -1)
    tabs = [np.load(tmp_path / ("table%d.npz" % r)) for r in range(world)]
    n_r = [int(t["n_rooms"]) for t in tabs]
    room_off = np.concatenate([[0], np.cumsum(n_r)])
    node_off = np.concatenate([[0], np.cumsum([len(t["room"]) for t in tabs])])
    if empty_rank >= 0:
        assert len(tabs[empty_rank]["room"]) == 0
    emb = np.concatenate([t["emb"].reshape(-1, 16) for t in tabs]).astype(np.float32)
    rooms = np.concatenate([t["room"] + room_off[r] for r, t in enumerate(tabs)]).astype(np.int32)
    T, lists = _queries(n_rooms_total=int(room_off[-1]))
    one = NodeIndex(emb.astype(np.float64), rooms, lib_=HmsgLib(PC.EMU_PATH))
    idx, room, score = one.query_objects(T, np.zeros(len(lists), np.int32), lists, 4)
    one.close()
    assert (idx >= 0).any()
    for r in range(world):
        z = np.load(tmp_path / ("ans%d.npz" % r))
        assert int(z["n"]) == len(rooms) and list(z["node_off"]) == list(node_off) and list(z["room_off"]) == list(room_off)
        assert np.array_equal(z["idx"], idx) and np.array_equal(z["room"], room) and np.array_equal(z["score"], score), r


def test_an_invalid_table_fails_on_every_rank(tmp_path):
    """rank 1 hands in n_rooms_local = 0 with nodes that have rooms: it must not return before the collectives and leave rank 0
    waiting -- the ranks agree first, and both calls fail"""
    _double()
    _spawn(_gather_worker, 2, str(tmp_path), -1, 1)
    r0, r1 = open(tmp_path / "res0.txt").read(), open(tmp_path / "res1.txt").read()
    assert r1.startswith("failed") and "outside n_rooms_local" in r1
    assert r0.startswith("failed") and "another rank" in r0


def _reduce_worker(rank, world, tmp):
    os.environ["HMSG_RCCL_LIB"] = DOUBLE
    from holoagent_amd._lib import Comm, HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    idp = os.path.join(tmp, "id.bin")
    if rank == 0:
        open(idp + ".tmp", "wb").write(Comm.unique_id(lib_=L))
        os.replace(idp + ".tmp", idp)
    comm = Comm.create(_wait_id(idp), rank, world, lib_=L)
    frames = _scene_frames(31, n_frames=4)
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=20, outlier_radius=0.3))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    a, b = rank * 2, rank * 2 + 2
    sc.set_frame_window(a)
    sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
    sc.fuse_frames()
    sc.allreduce_feature_sums(comm)
    sums, cnt = sc.feature_sums()
    feats, _ = sc.map_feats(counter=True)
    np.savez(os.path.join(tmp, "sum%d.npz" % rank), sums=sums, cnt=cnt, feats=feats)
    comm.close()
    sc.close()


def test_allreduce_feature_sums_with_two_ranks(tmp_path):
    from holoagent_amd._lib import HmsgLib
    _double()
    _spawn(_reduce_worker, 2, str(tmp_path))
    frames = _scene_frames(31, n_frames=4)
    S = PC.stack_frames(frames)
    sc = PC.make_scene(HmsgLib(PC.EMU_PATH), frames, dict(feat_dim=16, outlier_nb_points=20, outlier_radius=0.3))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sums, cnt = sc.feature_sums()
    feats, _ = sc.map_feats(counter=True)
    sc.close()
    a, b = np.load(tmp_path / "sum0.npz"), np.load(tmp_path / "sum1.npz")
    assert np.array_equal(a["sums"], b["sums"]) and np.array_equal(a["cnt"], b["cnt"])       # every rank holds the same totals
    assert np.array_equal(a["cnt"], cnt) and cnt.max() >= 2                                   # frame counters: exact
    np.testing.assert_allclose(a["sums"], sums, rtol=0, atol=1e-5)                            # float32 sums up to the order of addition
    np.testing.assert_allclose(a["feats"], feats, rtol=0, atol=1e-5)


# ---- configs[4] behind the C ABI only: hmsg_allreduce_feature_sums + hmsg_merge_tree_sharded (local levels, agreement, cross-rank
# joins over ncclSend / ncclRecv) -- no torch.distributed anywhere in the episode's path
def _episode_frames(n_frames):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=11, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48, n_frames=n_frames,
                     n_masks=8, feat_dim=16, yaw_step_deg=25.0)
    scn = SynthScene(spec)
    return [scn.frame(i) for i in range(n_frames)]


def _episode_scene(L, frames, window=None):
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, merge_type=1, outlier_nb_points=20, outlier_radius=0.3))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    a, b = window if window is not None else (0, len(frames))
    if window is not None:
        sc.set_frame_window(a)
    sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
    sc.fuse_frames()
    return sc


def _episode_worker(rank, world, tmp, n_frames, chunk):
    os.environ["HMSG_RCCL_LIB"] = DOUBLE
    from holoagent_amd._lib import Comm, HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    idp = os.path.join(tmp, "id.bin")
    if rank == 0:
        open(idp + ".tmp", "wb").write(Comm.unique_id(lib_=L))
        os.replace(idp + ".tmp", idp)
    comm = Comm.create(_wait_id(idp), rank, world, lib_=L)
    sc = _episode_scene(L, _episode_frames(n_frames), (rank * chunk, min(n_frames, (rank + 1) * chunk)))
    sc.allreduce_feature_sums(comm)
    holds = sc.merge_tree_sharded(comm, n_frames)
    assert holds == (rank == 0)
    if holds:
        inst = sc.instances()
        sc.pool_instances()
        np.savez(os.path.join(tmp, "episode.npz"), sizes=np.array([len(c) for c in inst]), pts=np.concatenate(inst) if inst else np.zeros((0, 3)),
                 pooled=sc.instance_feats())
    comm.close()
    sc.close()


@pytest.mark.parametrize("world,n_frames,chunk", [(2, 4, 2), (3, 5, 2)])
def test_sharded_merge_tree_behind_the_c_abi(tmp_path, world, n_frames, chunk):
    """two even ranks; three ranks with a shorter last window (the last rank stops its local tree below the others and is carried
    up): the root's instances are bit-identical to one process over all frames, the pooled features within 1e-5"""
    from holoagent_amd._lib import HmsgLib
    _double()
    _spawn(_episode_worker, world, str(tmp_path), n_frames, chunk)
    sc = _episode_scene(HmsgLib(PC.EMU_PATH), _episode_frames(n_frames))
    sc.merge_instances()
    ref = sc.instances()
    sc.pool_instances()
    ref_pooled = sc.instance_feats()
    sc.close()
    z = np.load(tmp_path / "episode.npz")
    assert len(ref) > 3 and z["sizes"].tolist() == [len(c) for c in ref]
    assert np.array_equal(z["pts"], np.concatenate(ref))
    np.testing.assert_allclose(z["pooled"], ref_pooled, rtol=0, atol=1e-5)


# ---- configs[3] through the graph object: hmsg_graph_allgather_index (node tables + the levels above them, global ids)
def _graph_worker(rank, world, tmp):
    os.environ["HMSG_RCCL_LIB"] = DOUBLE
    from holoagent_amd._lib import Comm, HmsgLib, SceneGraph
    L = HmsgLib(PC.EMU_PATH)
    idp = os.path.join(tmp, "id.bin")
    if rank == 0:
        open(idp + ".tmp", "wb").write(Comm.unique_id(lib_=L))
        os.replace(idp + ".tmp", idp)
    comm = Comm.create(_wait_id(idp), rank, world, lib_=L)
    frames = _scene_frames(40 + rank, n_frames=6)
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=20, outlier_radius=0.3, feat_dbscan_min=8))
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    cg = SceneGraph.build(sc, S["pose"], S["f_g"], num_views=3, host_threads=1)
    cnt = cg.counts()
    rooms = cg.rooms()
    rng = np.random.Generator(np.random.PCG64(100 + rank))
    names = rng.standard_normal((cnt["rooms"], 16))
    names /= np.linalg.norm(names, axis=1, keepdims=True)
    nodes, emb = sc.nodes(embeddings=True)
    np.savez(os.path.join(tmp, "graph%d.npz" % rank), emb=emb, room=np.array([int(n["room"]) for n in nodes], np.int32), names=names,
             floors=np.array([r["floor"] for r in rooms], np.int32), n_floors=cnt["floors"],
             keys=np.array([int(r["room_id"].split("_")[-1]) for r in rooms], np.int32),
             views=np.concatenate([np.asarray(cg.room_embeddings(i), np.float64).reshape(-1, 16) for i in range(cnt["rooms"])] + [np.zeros((0, 16))]),
             n_views=np.array([len(cg.room_embeddings(i)) for i in range(cnt["rooms"])], np.int64))
    ix, noff, roff, foff = cg.allgather_index(comm, names)
    T, Tr, fl = _hier_queries(int(foff[-1]))
    out = {}
    for mode in (1, 2, 0):
        sel, idx, room, score = ix.query_hier(T, np.zeros(len(T), np.int32), Tr, fl, np.full(len(T), mode, np.int32), 3, max_rooms=16)
        out["sel%d" % mode] = np.array([s + [-1] * (16 - len(s)) for s in sel], np.int32)
        out["idx%d" % mode], out["room%d" % mode], out["score%d" % mode] = idx, room, score
    np.savez(os.path.join(tmp, "gans%d.npz" % rank), noff=noff, roff=roff, foff=foff, **out)
    ix.close()
    cg.close()
    comm.close()
    sc.close()


def _hier_queries(n_floors_total, Q=8, D=16):
    rng = np.random.Generator(np.random.PCG64(3))
    T = rng.standard_normal((Q, 2, D)).astype(np.float32)
    Tr = rng.standard_normal((Q, D)).astype(np.float32)
    Tr /= np.linalg.norm(Tr, axis=1, keepdims=True)
    return T, Tr, (np.arange(Q) % max(n_floors_total, 1)).astype(np.int32)


@pytest.mark.parametrize("world", [2, 3])
def test_graph_allgather_index_with_several_ranks(tmp_path, world):
    """every rank's graph (its own scene: floors, rooms, KMeans views, objects) -> one index per rank; each answers hmsg_query_hier
    (label mode, view mode, no room stage) like ONE index built from the concatenated tables with shifted room / floor ids"""
    from holoagent_amd._lib import HmsgLib, NodeIndex
    _double()
    _spawn(_graph_worker, world, str(tmp_path))
    tabs = [np.load(tmp_path / ("graph%d.npz" % r)) for r in range(world)]
    roff = np.concatenate([[0], np.cumsum([len(t["floors"]) for t in tabs])])
    foff = np.concatenate([[0], np.cumsum([int(t["n_floors"]) for t in tabs])])
    noff = np.concatenate([[0], np.cumsum([len(t["room"]) for t in tabs])])
    emb = np.concatenate([t["emb"].reshape(-1, 16) for t in tabs]).astype(np.float32)
    rooms = np.concatenate([t["room"] + roff[r] for r, t in enumerate(tabs)]).astype(np.int32)
    floor_rooms = []
    for r, t in enumerate(tabs):
        for f in range(int(t["n_floors"])):
            floor_rooms.append([int(roff[r] + i) for i in np.where(t["floors"] == f)[0]])
    views = []
    for t in tabs:
        o = np.concatenate([[0], np.cumsum(t["n_views"])])
        views += [t["views"][o[i]:o[i + 1]] for i in range(len(t["n_views"]))]
    one = NodeIndex(emb.astype(np.float64), rooms, lib_=HmsgLib(PC.EMU_PATH))
    one.set_hierarchy(floor_rooms, np.concatenate([t["names"] for t in tabs]), views, np.concatenate([t["keys"] for t in tabs]).tolist())
    T, Tr, fl = _hier_queries(int(foff[-1]))
    assert len(emb) >= 2 * world and int(foff[-1]) == world
    for mode in (1, 2, 0):
        sel, idx, room, score = one.query_hier(T, np.zeros(len(T), np.int32), Tr, fl, np.full(len(T), mode, np.int32), 3, max_rooms=16)
        sel = np.array([s + [-1] * (16 - len(s)) for s in sel], np.int32)
        assert (idx >= 0).any()
        for r in range(world):
            z = np.load(tmp_path / ("gans%d.npz" % r))
            assert list(z["noff"]) == list(noff) and list(z["roff"]) == list(roff) and list(z["foff"]) == list(foff)
            assert np.array_equal(z["sel%d" % mode], sel) and np.array_equal(z["idx%d" % mode], idx), (mode, r)
            assert np.array_equal(z["room%d" % mode], room) and np.array_equal(z["score%d" % mode], score), (mode, r)
    one.close()

