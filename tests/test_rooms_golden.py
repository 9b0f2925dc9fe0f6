"""A9 (rooms level of the hierarchy) and A11 (graph edges) against the REFERENCE's own functions:
compute_room_embeddings (utils/graph_utils.py:192-356, tests/golden/roomemb.npz) and create_graph_new /
load_hmsg_graph (graph.py:1752-1775, 1892-1987, tests/golden/graphedges.json)."""
import json
import os

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC


def _lib(gpu):
    from holoagent_amd._lib import HmsgLib
    if gpu:
        return HmsgLib()
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    return HmsgLib(PC.EMU_PATH)


def check_room_embeddings(L):
    from holoagent_amd.graph import compute_room_embeddings
    from oracle.refdrive.gen_golden import roomemb_case
    z = GI.load("roomemb")
    rooms, poses, embs, pmin, pmax = roomemb_case()
    repr_embs, repr_ids, r2i, clip = compute_room_embeddings(rooms, poses, embs, pmin, pmax, 24, None, lib=L)
    assert len(repr_ids) == int(z["n_rooms"]) == len(r2i)
    for i in range(len(rooms)):
        assert [int(v) for v in r2i[i]] == z["img_ids_%d" % i].tolist(), i
        assert [int(v) for v in repr_ids[i]] == z["repr_ids_%d" % i].tolist(), i
        np.testing.assert_array_equal(np.array(repr_embs[i], np.float32).reshape(len(repr_ids[i]), -1), z["repr_embs_%d" % i])
        np.testing.assert_array_equal(np.asarray(clip[i], np.float32), z["clip_%d" % i])


def test_room_embeddings_match_reference_emu():
    check_room_embeddings(_lib(False))


@pytest.mark.gpu
def test_room_embeddings_match_reference_gpu():
    check_room_embeddings(_lib(True))


def test_min_dist_kernel_equals_cdist():
    from scipy.spatial import distance
    from holoagent_amd._lib import points_min_dist_2d
    L = _lib(False)
    rng = np.random.Generator(np.random.PCG64(4))
    sets = [rng.uniform(-5, 5, (n, 2)) for n in (1, 700, 0, 33)]
    q = rng.uniform(-6, 6, (57, 2))
    got = points_min_dist_2d(sets, q, lib_=L)
    for s, pts in enumerate(sets):
        ref = np.min(distance.cdist(q, pts, metric="euclidean"), axis=1) if len(pts) else np.full(len(q), np.inf)
        assert np.array_equal(got[:, s], ref)


def _node_key(n):
    from holoagent_amd.graph import Floor, Object, Room, View
    for cls, tag, attr in ((Floor, "floor", "floor_id"), (Room, "room", "room_id"), (Object, "object", "object_id"),
                           (View, "view", "view_id")):
        if isinstance(n, cls):
            return "%s:%s" % (tag, getattr(n, attr))
    return "root:%s" % n


def test_graph_edges_match_reference(tmp_path):
    """create_graph_new on a freshly built graph (no Room - View edge: the View carries the int room index) and the
    edges load_hmsg_graph adds, against the reference's own."""
    from holoagent_amd.graph import Floor, Graph, Object, Room, View, _Pcd
    from oracle.refdrive.gen_golden import build_persist_graph, persist_case
    ref = json.load(open(os.path.join(GI.GOLDEN, "graphedges.json")))
    L = _lib(False)
    fl, rooms, objects, views = build_persist_graph(persist_case(), Floor, Room, Object, View, lambda p: _Pcd(p))
    for v in views:
        v.room_id = int(str(v.room_id).split("_")[-1])
    g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=8))), lib=L)
    g.floors, g.rooms, g.objects, g.views = [fl], rooms, objects, views
    g.create_graph_new()
    edges = lambda gr: sorted(sorted([_node_key(a), _node_key(b)]) for a, b in gr.graph.edges())
    assert edges(g) == ref["built"]
    g.save_hmsg_graph(str(tmp_path))
    g2 = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=8))), lib=L)
    g2.load_hmsg_graph(str(tmp_path))
    assert edges(g2) == ref["loaded"]
    assert [o.object_id for o in g2.objects] == ref["object_order"]
    assert [v.view_id for v in g2.views] == ref["view_order"]


def test_stage_artefacts_round_trip(tmp_path):
    """save_full_pcd / save_full_pcd_feats (.pt, graph.py:3797-3830) / save_masked_pcds and their loaders."""
    import torch
    from holoagent_amd.graph import Graph, _Pcd
    L = _lib(False)
    rng = np.random.Generator(np.random.PCG64(9))
    g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=6))), lib=L)
    g.full_pcd = _Pcd(rng.uniform(0, 1, (50, 3)))
    g.full_feats_array = rng.standard_normal((50, 6)).astype(np.float32)
    g.mask_pcds = [_Pcd(rng.uniform(0, 1, (n, 3))) for n in (12, 3, 40, 0)]
    g.mask_feats = [rng.standard_normal(6).astype(np.float32) for _ in range(4)]
    feats0 = np.array(g.mask_feats)
    g.save_masked_pcds(str(tmp_path), state="both")          # drops the clouds with fewer than 10 points
    assert len(g.mask_pcds) == 2 and sorted(os.listdir(tmp_path / "objects")) == ["pcd_0.ply", "pcd_1.ply"]
    g.save_full_pcd(str(tmp_path))
    g.save_full_pcd_feats(str(tmp_path))
    t = torch.load(tmp_path / "mask_feats.pt")
    assert t.dtype == torch.float32 and np.array_equal(t.numpy(), feats0[[0, 2]])
    h = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=6))), lib=L)
    assert h.load_full_pcd(str(tmp_path / "nowhere")) is None
    assert np.array_equal(h.load_full_pcd(str(tmp_path)).points, g.full_pcd.points)
    mf = h.load_full_pcd_feats(str(tmp_path), normalize=False)
    assert np.array_equal(mf, feats0[[0, 2]])
    ff = h.load_full_pcd_feats(str(tmp_path), full_feats=True, normalize=True)
    np.testing.assert_allclose(np.linalg.norm(ff, axis=1), 1.0, atol=1e-6)
    clouds = h.load_masked_pcds_new(str(tmp_path))
    assert len(clouds) == 2 and np.array_equal(clouds[1].points, g.mask_pcds[1].points)


@pytest.mark.gpu
def test_graph_build_with_room_regions_gpu(tmp_path):
    """the same on the MI355X: View nodes, the save / load round trip, rank_goal_views (graph.py:2864-2897) and the
    string-instruction driver on the HIP library"""
    check_graph_build_with_room_regions(tmp_path, _lib(True))


def test_graph_build_with_room_regions(tmp_path):
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    check_graph_build_with_room_regions(tmp_path, _lib(False))


def check_graph_build_with_room_regions(tmp_path, L):
    """The mirrored segment_hmsg_room (graph.py:1073-1189 from the rooms' 2-D regions on): room clouds, representative
    view embeddings, View nodes with the reference's id scheme, and the Room - View edge quirk through a save / load."""
    from holoagent_amd.graph import Graph, Room, View
    from tests.graph_fixture import SynthDataset, SynthEncoders, tiny_scene
    scn = tiny_scene(6, 32)
    ds = SynthDataset(scn)
    enc = SynthEncoders(ds, ["background", "wall", "office", "kitchen", "chair", "table"])
    cfg = dict(main=dict(device_id=0), models=dict(clip=dict(type="ViT-B/32", feat_dim=32)),
               pipeline=dict(voxel_size=0.05, skip_frames=1, merge_type="sequential", max_masks=8))
    g = Graph(cfg, dataset=ds, encoders=enc, lib=L)
    g.create_feature_map()
    assert len(g._view_feats) == 6
    lo, hi = scn.rooms[0]
    mid = (lo[0] + hi[0]) / 2
    regions = [[np.array([[x, z] for x in np.arange(lo[0], mid, 0.1) for z in np.arange(lo[2], hi[2], 0.1)]),
                np.array([[x, z] for x in np.arange(mid, hi[0], 0.1) for z in np.arange(lo[2], hi[2], 0.1)])]]
    g.set_label_feats(enc.encode_text(["chair", "table"]), ["chair", "table"])
    g.build_hier_multimodal_scene_graph(str(tmp_path), room_regions=regions)
    assert [r.room_id for r in g.rooms] == ["0_0", "0_1"]
    assert sum(len(r.sample_images) for r in g.rooms) >= 6 and all(len(r.embeddings) >= 1 for r in g.rooms)
    assert all(len(r.pcd.points) > 0 for r in g.rooms)
    assert [v.view_id for v in g.views] == ["0_%d_%d" % (v.room_id, k) for k, v in enumerate(g.views)]
    assert all(isinstance(v.room_id, int) for v in g.views)
    rv = lambda gr: [(a, b) for a, b in gr.graph.edges() if {type(a), type(b)} == {Room, View}]
    assert len(rv(g)) == 0                                  # build time: string room id vs int view.room_id never match
    g2 = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=32))), encoders=enc, lib=L)
    g2.load_hmsg_graph(str(tmp_path / "graph"))
    assert len(rv(g2)) == len(g.views) and len(g2.objects) == len(g.objects)
    # slow-path view search (graph.py:2864-2897): top views of the candidate rooms for an object query
    best, top, sims = g2.rank_goal_views("chair", g2.rooms, top_k=24)
    allv = [i for r in g2.rooms for i in r.sample_images]
    ref = np.dot(g2.get_text_feats_multiple_templates(["chair"])[0].astype(np.float64),
                 np.stack([np.asarray(e, np.float64).reshape(-1) for r in g2.rooms for e in r.clip_embeddings]).T)
    np.testing.assert_allclose(sims, ref, rtol=0, atol=1e-12)
    assert best == allv[int(np.argmax(ref))] and top == [allv[int(i)] for i in np.argsort(ref)[-min(24, len(ref)):][::-1]]
    g2.generate_room_names(default_room_types=["office", "kitchen"])
    fl, rooms, objs, res = g2.query_hierarchy_protected_icra("find the chair in the %s" % g2.rooms[0].name, top_k=2)
    assert res["object_query"] == "chair" and len(objs) <= 2


def check_room_clouds_device(L):
    """A9's room clouds (graph.py:1086-1108) behind the C ABI (hmsg_room_clouds: nearest neighbours in the storey's slab of
    the map on the device, bit-equal ties by the restated cKDTree) == the host mirror (scipy cKDTree over the floor cloud,
    Open3D transform / select_by_index restated line by line), point for point, on a two-storey scene with three rooms."""
    from holoagent_amd.graph import Graph
    from holoagent_amd.synth import SceneSpec, SynthScene
    from scipy.spatial import cKDTree
    frames = []
    for fl in range(2):
        spec = SceneSpec(seed=40 + fl, rooms_x=2, rooms_z=1, room_size=(3.2, 2.6, 3.0), objects_per_room=3, width=96, height=72,
                         n_frames=6, n_masks=6, feat_dim=16, yaw_step_deg=40.0)
        scn = SynthScene(spec)
        for i in range(spec.n_frames):
            fr = scn.frame(i)
            fr["pose"] = np.array(fr["pose"], np.float64)
            fr["pose"][1, 3] += fl * 2.6
            frames.append(fr)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=40, outlier_radius=0.5))
    S = PC.stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=16))), lib=L)
    g.scene = sc
    from holoagent_amd.graph import _Pcd
    g.full_pcd = _Pcd(sc.map_points())
    g.segment_floors_manually(None)
    assert len(g.floors) >= 2
    checked = 0
    for floor in g.floors:
        fp = np.asarray(floor.pcd.points)
        if len(fp) < 500:
            continue
        lo, hi = fp[:, [0, 2]].min(0), fp[:, [0, 2]].max(0)
        mid = (lo[0] + hi[0]) / 2
        regions = []
        for x0, x1 in ((lo[0], mid), (mid, hi[0])):               # two rooms on a 5 cm lattice, one of them ragged
            xs, zs = np.arange(x0 + 0.1, x1 - 0.1, 0.05), np.arange(lo[1] + 0.1, hi[1] - 0.1, 0.05)
            cells = np.stack(np.meshgrid(xs, zs, indexing="ij"), -1).reshape(-1, 2)
            regions.append(cells[(np.arange(len(cells)) % 7) != 3])
        dev = g._room_clouds_device(floor, regions)
        assert dev is not None
        tree = cKDTree(fp)
        for r, cells in enumerate(regions):
            ref = np.asarray(g._room_cloud(floor, tree, cells).points)
            got = np.asarray(dev[r].points)
            assert got.shape == ref.shape and np.array_equal(got, ref), (floor.floor_id, r, got.shape, ref.shape)
            assert len(ref) > 100
            checked += 1
    assert checked >= 4
    sc.close()


def test_room_clouds_device_equals_host_emu():
    check_room_clouds_device(_lib(False))


@pytest.mark.gpu
def test_room_clouds_device_equals_host_gpu():
    check_room_clouds_device(_lib(True))
