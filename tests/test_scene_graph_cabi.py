"""(b) the graph as ONE object behind the C ABI (include/hmsg.h: hmsg_build_graph / hmsg_graph_begin + _finish, hmsg_save, hmsg_load,
hmsg_graph_query; holoagent_amd/csrc/hmsg_scene_graph.hip) against the Python mirror of the reference's Graph
(holoagent_amd/graph.py: build_hier_multimodal_scene_graph, save_hmsg_graph, load_hmsg_graph, query_hierarchy_batch -- itself pinned to
the reference's runs by tests/test_persist_golden.py, test_rooms_from_frames.py, test_objects_golden.py, test_graph_bookkeeping_cabi.py):

  * the directory hmsg_save writes is BYTE for byte the directory the mirror's save_hmsg_graph writes (floors / rooms / objects / views,
    .json and .ply) -- with the KMeans of the room level restated in C++ (hmsg_kmeans) against scikit-learn's in the mirror;
  * ids, names, lists and edges of the built graph (hmsg_graph_to_json) are the mirror's;
  * hmsg_load of that directory gives the nodes load_hmsg_graph gives, and hmsg_graph_query on it answers like the mirror's
    query_hierarchy_batch on the loaded graph."""
import json
import os

import numpy as np
import pytest

from tests import parity_common as PC


def _build(L, device, storeys=1):
    import dataclasses
    import torch
    import bench
    from holoagent_amd._lib import Scene
    from holoagent_amd.synth import SceneSpec
    spec = SceneSpec(seed=1234, n_frames=12, feat_dim=16, n_masks=32, width=96, height=72, rooms_x=2, rooms_z=1, room_size=(3.2, 2.6, 3.0),
                     yaw_step_deg=36.0, objects_per_room=3)
    inp = bench.build_scene_inputs(L, spec, device, torch)
    if storeys > 1:
        # one-storey scenes stacked 3.4 m apart (up = +y), their frames one after the other: segment_floors_manually has to split the
        # height, every storey gets its own room level (regions, room clouds, camera -> room table, KMeans) and its own ids
        parts = [inp]
        for fl in range(1, storeys):
            more = bench.build_scene_inputs(L, dataclasses.replace(spec, seed=spec.seed + fl), device, torch)
            pose = np.array(more["pose"], np.float64).reshape(-1, 4, 4)
            pose[:, 1, 3] += 3.4 * fl
            more["pose"] = np.ascontiguousarray(pose.reshape(-1, 16))
            parts.append(more)
        inp = dict(inp)
        for k in ("rgb", "depth", "masks", "f_g", "f_masked", "f_crop"):
            inp[k] = torch.cat([q[k] for q in parts], 0).contiguous()
        inp["pose"] = np.ascontiguousarray(np.concatenate([q["pose"] for q in parts], 0))
        spec = dataclasses.replace(spec, n_frames=spec.n_frames * storeys)
    sc = Scene(lib_=L, device_id=0, height=spec.height, width=spec.width, max_frames=spec.n_frames, max_masks=32, feat_dim=spec.feat_dim)
    sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
    sc.finalize_map()
    return spec, inp, sc


def _rest(sc, inp):
    sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()


def check_graph_object(L, device, tmp_path, storeys=1, merge=False):
    from holoagent_amd._lib import SceneGraph
    from holoagent_amd.graph import Graph
    spec, inp, sc = _build(L, device, storeys)
    F, D = spec.n_frames, spec.feat_dim
    poses = [np.asarray(inp["pose"][i], np.float64).reshape(4, 4) for i in range(F)]
    fg = inp["f_g"].cpu().numpy()
    rng = np.random.Generator(np.random.PCG64(99))
    label_feats = rng.standard_normal((9, D)).astype(np.float32)
    label_feats /= np.linalg.norm(label_feats, axis=1, keepdims=True)
    label_names = ["label%d" % i for i in range(8)] + ["café \"table\""]              # (an id the JSON writer has to escape)
    if merge:
        # pipeline.merge_objects_graph (graph.py:2053-2058): ONE name for every object, so that every pair of a room's objects is a
        # candidate and the ones whose clouds touch are fused (Room.merge_objects) -- in the C graph by hmsg_graph_params::merge_objects_graph
        label_feats, label_names = label_feats[:1], ["thing"]
    blank = np.broadcast_to(np.zeros((), np.uint8), (spec.height, spec.width, 3))

    class FrameSource:
        frameId2imgPath = ["img/%05d.png" % i for i in range(F)]

        def __len__(self):
            return F

        def __getitem__(self, i):
            return blank, None, poses[i], None, None

        def get_camera_intrinsics(self):
            return inp["K"]
    # ---- the C graph: room level right after the map (KMeans on its host threads), the rest after the pooling
    cg = SceneGraph.begin(sc, np.stack(poses), fg, poses_inv=np.stack([np.linalg.inv(p) for p in poses]), img_paths=FrameSource.frameId2imgPath,
                          num_views=5, host_threads=2, merge_objects_graph=1 if merge else 0)
    _rest(sc, inp)
    cg.finish(label_feats, label_names)
    # ---- the mirror on the same handle (num_views = 5 like the C graph: the scene has 12 frames)
    import holoagent_amd.graph as G
    full_cfg = dict(main=dict(device_id=0), models=dict(clip=dict(feat_dim=D)),
                    pipeline=dict(grid_resolution=0.05, skip_frames=1, views_on_device=True, merge_objects_graph=bool(merge)))
    g = Graph.from_scene(sc, cfg=full_cfg, lib=L)
    g.dataset = FrameSource()
    g._poses = poses
    g.set_view_feats(fg)
    g.set_label_feats(label_feats, label_names)
    orig, orig_pick = G.compute_room_embeddings, G._closest_member
    G.compute_room_embeddings = lambda *a, **k: orig(a[0], a[1], a[2], a[3], a[4], 5, *a[6:], **k)
    # (a two-member cluster is an exact tie that the reference decides inside BLAS: both sides take the library's rule here --
    #  products accumulated in float64, first maximum -- so that the files can be compared byte for byte)
    G._closest_member = lambda cluster, centre: int(np.argmax(np.asarray(cluster, np.float64) @ np.asarray(centre, np.float64)))
    try:
        g.build_hier_multimodal_scene_graph(str(tmp_path / "py"))
    finally:
        G.compute_room_embeddings, G._closest_member = orig, orig_pick
    cnt = cg.counts()
    assert (cnt["floors"], cnt["rooms"], cnt["views"], cnt["objects"]) == (len(g.floors), len(g.rooms), len(g.views), len(g.objects))
    assert cnt["rooms"] >= 1 and cnt["objects"] >= 3 and cnt["view_object_links"] >= 3
    if merge:
        assert len(g.objects) < len(sc.nodes()), "no pair of objects was merged: the test scene does not exercise Room.merge_objects"
    # (every camera lands in one room of its storey; a room nobody stands in borrows its nearest camera: graph_utils.py:280-291)
    assert cnt["views"] == F if storeys == 1 else (cnt["views"] >= F and cnt["floors"] >= storeys and cnt["rooms"] >= storeys)
    assert any(len(r.sample_images) >= 5 for r in g.rooms), "no room went through KMeans"
    # ---- topology
    d = cg.to_dict()
    assert [f["floor_id"] for f in d["floors"]] == [f.floor_id for f in g.floors]
    assert [f["rooms"] for f in d["floors"]] == [[r.room_id for r in f.rooms] for f in g.floors]
    for a, r in zip(d["rooms"], g.rooms):
        assert (a["room_id"], a["name"], a["floor_id"]) == (r.room_id, r.name, r.floor_id)
        assert a["objects"] == [o.object_id for o in r.objects] and a["views"] == [v.view_id for v in r.views]
        assert a["represent_images"] == list(r.represent_images) and a["sample_images"] == list(r.sample_images)
        assert a["n_points"] == len(np.asarray(r.pcd.points)) and a["n_vertices"] == len(r.vertices) and a["n_embeddings"] == len(r.embeddings)
    for a, v in zip(d["views"], g.views):
        assert (a["view_id"], a["room_id"], a["img_id"], a["img_path"]) == (v.view_id, v.room_id, v.img_id, v.img_path)
        assert a["object_ids"] == v.object_ids and a["text_discription"] == v.text_discription
    for a, o in zip(d["objects"], g.objects):
        assert (a["object_id"], a["room_id"], a["name"], a["best_view_id"]) == (o.object_id, o.room_id, o.name, o.best_view_id)
        assert a["view_ids"] == o.view_ids and a["instance"] == o._instance
    # edges = create_graph_new's, in its insertion order (node ids: 0 building, floors, rooms, objects, views)
    nid = {id(n): 1 + k for k, n in enumerate(g.floors + g.rooms + g.objects + g.views)}
    idof = lambda n: 0 if isinstance(n, int) else nid[id(n)]
    canon = lambda pairs: sorted(tuple(sorted(p)) for p in pairs)
    got_e = [tuple(e) for e in cg.edges().tolist()]
    assert canon(got_e) == canon((idof(a), idof(b)) for a, b in g.graph.edges())          # (networkx hands edges back node by node)
    want = []
    for f in g.floors:                                                                   # ... and create_graph_new's insertion order
        want.append((0, idof(f)))
        for r in f.rooms:
            want.append((idof(f), idof(r)))
            want += [(idof(r), idof(o)) for o in r.objects]
    for v in g.views:
        want += [(idof(v), idof(o)) for o in g.objects if o.object_id in v.object_ids]
    assert got_e == want
    # ---- persistence: byte for byte
    cg.save(tmp_path / "c")
    for sub in ("floors", "rooms", "objects", "views"):
        a, b = sorted(os.listdir(tmp_path / "py" / "graph" / sub)), sorted(os.listdir(tmp_path / "c" / sub))
        assert a == b and len(a) > 0, sub
        for f in a:
            assert open(tmp_path / "py" / "graph" / sub / f, "rb").read() == open(tmp_path / "c" / sub / f, "rb").read(), (sub, f)
    # ---- load + query
    lg = SceneGraph.load(tmp_path / "c", lib_=L)
    g2 = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=D))), lib=L)
    g2.load_hmsg_graph(str(tmp_path / "py" / "graph"))
    ld = lg.to_dict()
    assert [r["room_id"] for r in ld["rooms"]] == [r.room_id for r in g2.rooms]
    assert [o["object_id"] for o in ld["objects"]] == [o.object_id for o in g2.objects]
    assert [o["name"] for o in ld["objects"]] == [o.name for o in g2.objects]
    assert [v["view_id"] for v in ld["views"]] == [v.view_id for v in g2.views]
    assert [r["views"] for r in ld["rooms"]] == [list(r.views) for r in g2.rooms]
    nid2 = {id(n): 1 + k for k, n in enumerate(g2.floors + g2.rooms + g2.objects + g2.views)}
    idof2 = lambda n: 0 if isinstance(n, int) else nid2[id(n)]
    assert canon(tuple(e) for e in lg.edges().tolist()) == canon((idof2(a), idof2(b)) for a, b in g2.graph.edges())
    Q = 6
    rng = np.random.default_rng(5)
    T = rng.standard_normal((Q, 2, D)).astype(np.float32)
    T /= np.linalg.norm(T, axis=-1, keepdims=True)
    room_names = rng.standard_normal((len(g2.rooms), D))
    room_names /= np.linalg.norm(room_names, axis=1, keepdims=True)
    Tr = np.ascontiguousarray(room_names[rng.integers(0, len(g2.rooms), Q)], np.float32)
    zero = np.zeros(Q, np.int32)
    for mode in (1, 2, 0):
        sel, idx, room, score = lg.query(T, zero, Tr, zero - 1, zero + mode, 3, room_name_emb=room_names)
        ix = g2._node_index()
        gl = {r.room_id: i for i, r in enumerate(g2.rooms)}
        ix.set_hierarchy([[gl[r.room_id] for r in f.rooms] for f in g2.floors], room_names,
                         [np.stack(r.embeddings) if len(r.embeddings) else np.zeros((0, D)) for r in g2.rooms],
                         [int(str(r.room_id).split("_")[-1]) for r in g2.rooms])
        sel2, idx2, room2, score2 = ix.query_hier(T, zero, Tr, zero - 1, zero + mode, 3)
        assert sel == sel2 and np.array_equal(idx, idx2) and np.array_equal(room, room2) and np.array_equal(score, score2), mode
    # the built graph answers on the resident table too (float32 pooled features: the scores of the freshly built graph)
    sel, idx, room, score = cg.query(T, zero, Tr, zero - 1, zero + 1, 3, room_name_emb=np.ascontiguousarray(room_names))
    assert (idx >= 0).any()
    lg.close()
    cg.close()
    sc.close()


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_graph_object_equals_the_mirror_on_the_simulator(tmp_path):
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(PC.EMU_PATH), torch.device("cpu"), tmp_path)


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_graph_object_two_storeys_on_the_simulator(tmp_path):
    """The same comparison on a two-storey scene: the room level runs once per storey inside hmsg_graph_begin (the resident room state of
    one storey must not leak into the next), ids carry the storey, the loaded graph answers per floor."""
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(PC.EMU_PATH), torch.device("cpu"), tmp_path, storeys=2)


@pytest.mark.gpu
def test_graph_object_two_storeys_gpu(tmp_path):
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(), torch.device("cuda", 0), tmp_path, storeys=2)


@pytest.mark.gpu
def test_graph_object_equals_the_mirror_gpu(tmp_path):
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(), torch.device("cuda", 0), tmp_path)


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_graph_object_merges_objects_like_the_mirror_on_the_simulator(tmp_path):
    """pipeline.merge_objects_graph: Room.merge_objects inside hmsg_graph_finish (same-name overlap tests on the device, the
    reference's chaining, ids re-numbered, merged clouds / mean embeddings saved) against the mirror with the same switch:
    nodes, edges in order, saved directories byte for byte, the loaded graph's answers."""
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(PC.EMU_PATH), torch.device("cpu"), tmp_path, merge=True)


@pytest.mark.gpu
def test_graph_object_merges_objects_like_the_mirror_gpu(tmp_path):
    import torch
    from holoagent_amd._lib import HmsgLib
    check_graph_object(HmsgLib(), torch.device("cuda", 0), tmp_path, merge=True)

