"""Development tests: the product's HIP sources compiled for the host kernel simulator (tests/emu),
driven through the same C ABI and compared with the oracle.  Skipped when the simulator library has
not been built (make -C holoagent_amd/csrc emu)."""
import os

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


@pytest.fixture(scope="module")
def L():
    from holoagent_amd._lib import HmsgLib
    return HmsgLib(PC.EMU_PATH)


def test_map_and_fuse_small(L):
    z = GI.load("build_hier")
    frames = GI.unpack_frames(z)[:6]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300            # few low-res frames: keep a useful part of the cloud
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    assert 0 < ref_pts.shape[0] < sc.map_size_unfiltered()
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    sc.close()


def test_four_word_mask_bitsets(L):
    """Up to 256 masks per frame (4-word bitsets per pixel, NW = 4): frames whose masks are re-cut into 150-200 small
    overlapping patches; map, per-pixel fusion and the 3-D masks still equal the oracle."""
    z = GI.load("build_hier")
    frames = GI.unpack_frames(z)[:3]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300
    rng = np.random.Generator(np.random.PCG64(77))
    D = cfg["feat_dim"]
    for k, f in enumerate(frames):
        H, W = f["depth"].shape
        M = 150 + 25 * k                                  # 150, 175, 200: three and four 64-bit words
        masks = np.zeros((M, H, W), bool)
        for m in range(M):
            h, w = int(rng.integers(3, H // 3)), int(rng.integers(3, W // 3))
            y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
            masks[m, y:y + h, x:x + w] = True
        unit = lambda a: (a / np.linalg.norm(a, axis=-1, keepdims=True)).astype(np.float32)
        f["masks"] = masks
        f["f_masked"] = unit(rng.standard_normal((M, D)))
        f["f_crop"] = unit(rng.standard_normal((M, D)))
    sc = PC.make_scene(L, frames, dict(feat_dim=D, outlier_nb_points=300))
    assert sc.cfg.max_masks == 200
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    sc.close()


def test_frames_without_masks_or_depth(L):
    """Edge inputs inside an episode: a frame SAM found nothing in (zero masks) and a frame without a single valid depth
    pixel; the build goes through and map, fusion and 3-D masks equal the oracle."""
    z = GI.load("build_hier")
    frames = GI.unpack_frames(z)[:5]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300
    D = cfg["feat_dim"]
    H, W = frames[0]["depth"].shape
    frames[1]["masks"] = np.zeros((0, H, W), bool)
    frames[1]["f_masked"] = np.zeros((0, D), np.float32)
    frames[1]["f_crop"] = np.zeros((0, D), np.float32)
    frames[3]["depth"] = np.zeros_like(frames[3]["depth"])
    sc = PC.make_scene(L, frames, dict(feat_dim=D, outlier_nb_points=300))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    assert sc.frame_num_masks(1) == 0 and all(len(m) == 0 for m in sc.frame_masks3d(3))
    sc.close()


def test_mask_walk_heavy_voxel_path(L):
    """The wave-per-voxel replay of heavy mask voxels (k_mwalk_heavy: bulk integer additions inside a binade) forced
    on every voxel: still bit-identical."""
    z = GI.load("build_ragged")
    frames = GI.unpack_frames(z)[:3]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300
    os.environ["HMSG_DEBUG_MWALK_HEAVY"] = "1"
    try:
        sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300))
        S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
        PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
        sc.close()
    finally:
        os.environ.pop("HMSG_DEBUG_MWALK_HEAVY", None)


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="slow on the simulator (minutes); covered on the GPU")
@pytest.mark.parametrize("merge_type", ["sequential", "hierarchical"])
def test_merge_and_pool_small(L, merge_type):
    z = GI.load("build_hier")
    frames = GI.unpack_frames(z)[:12]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300
    cfg["merge_type"] = merge_type
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300,
                                       merge_type=1 if merge_type == "hierarchical" else 0, feat_dbscan_min=20))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    ref_feats, _ = PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks=False)
    import oracle.hmsg_oracle as O
    # smaller min_samples so the cosine DBSCAN actually forms clusters on this tiny scene
    orig = O.feats_denoise_dbscan
    O.feats_denoise_dbscan = lambda f, eps=0.01, min_points=100: orig(f, eps=0.01, min_points=20)
    try:
        got, feats = PC.check_merge_pool(sc, frames, cfg, ref_pts, ref_feats)
    finally:
        O.feats_denoise_dbscan = orig
    assert len(got) > 3
    print("tie queries answered by the cKDTree replay:", sc.num_tie_queries())
    sc.close()


def test_query_golden(L):
    PC.check_query_golden(L)


def test_graph_end_to_end_tiny(L, tmp_path):
    """holoagent_amd.graph.Graph (the mirror of the reference's Graph) end to end: build, assemble, save in the
    reference's on-disk layout, load, query; retrieval equals the numpy restatement on the loaded table."""
    from holoagent_amd.graph import Graph
    from oracle import hmsg_oracle as O
    from tests.graph_fixture import SynthDataset, SynthEncoders, tiny_scene
    scn = tiny_scene(4, 32)
    ds = SynthDataset(scn)
    enc = SynthEncoders(ds, ["background", "wall", "office", "kitchen", "chair", "table"])
    cfg = dict(main=dict(device_id=0), models=dict(clip=dict(type="ViT-B/32", feat_dim=32)),
               pipeline=dict(voxel_size=0.05, skip_frames=1, merge_type="sequential", max_masks=8))
    g = Graph(cfg, dataset=ds, encoders=enc, lib=L)
    g.create_feature_map()
    # tiny scene: the literal outlier filter (1000 neighbours in 1 m) would delete everything -> rebuilt map
    assert len(g.mask_feats) == len(g.mask_pcds)
    g.set_label_feats(enc.encode_text(["chair", "table"]), ["chair", "table"])
    lo, hi = scn.rooms[0]
    rooms = [dict(floor=0, name="office", vertices=[[x, z] for x in np.arange(lo[0], hi[0], 0.1) for z in np.arange(lo[2], hi[2], 0.1)],
                  view_frames=[0, 2], view_embeddings=[enc.encode_text(["office"])[0], enc.encode_text(["kitchen"])[0]])]
    g.build_hier_multimodal_scene_graph(str(tmp_path), rooms=rooms)
    assert len(g.floors) >= 1
    # A10 room association on the device == find_intersection_share (utils/graph_utils.py:160-189) on the host
    from holoagent_amd.graph import find_intersection_share
    share = g.scene.instance_room_share([r.vertices for r in g.rooms], 0.2)
    for i, pts in enumerate(g.scene.instances()):
        for k, r in enumerate(g.rooms):
            ref = find_intersection_share(r.vertices, pts[:, [0, 2]], 0.2) if len(pts) else 0
            assert abs(share[i, k] - ref) < 1e-12, (i, k, share[i, k], ref)
    # the node table behind the C ABI (hmsg_get_nodes / hmsg_index_from_nodes) is the object list
    nodes, emb = g.scene.nodes(embeddings=True)
    assert len(nodes) == len(g.objects)
    for n, e, o in zip(nodes, emb, g.objects):
        assert o.object_id == "%s_%d" % (g.rooms[int(n["room"])].room_id, int(n["counter"]))
        np.testing.assert_array_equal(e, np.asarray(o.embedding, np.float32))
    if len(nodes):
        ixn = g.scene.index_from_nodes()
        Tq = g.get_text_feats_multiple_templates(["chair", "background"])[None]
        a = ixn.query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
        b = g._node_index().query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
        # the exchange step of the multi-GPU scene mode behind the C ABI (hmsg_allgather_nodes) with one rank and no
        # communicator: the "global" table is the local one, offsets [0, n] / [0, rooms]
        from holoagent_amd._lib import Comm
        cm = Comm.single(lib_=L)
        ixg, noff, roff = g.scene.allgather_nodes(cm, len(g.rooms))
        assert list(noff) == [0, len(nodes)] and list(roff) == [0, len(g.rooms)] and ixg.N == len(nodes)
        c = ixg.query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
        assert np.array_equal(a[0], c[0]) and np.array_equal(a[1], c[1]) and np.array_equal(a[2], c[2])
        ixg.close()
        cm.close()
        ixn.close()
    # N2: the objects were written by the library's bulk writer (hmsg_save_objects): byte-identical with Object.save
    import os
    ref_dir = tmp_path / "objects_py"
    ref_dir.mkdir()
    for o in g.objects:
        assert o._instance is not None
        o.save(str(ref_dir))
    names = sorted(os.listdir(ref_dir))
    assert len(names) == 2 * len(g.objects) and names == sorted(os.listdir(tmp_path / "graph" / "objects"))
    for f in names:
        assert open(ref_dir / f, "rb").read() == open(tmp_path / "graph" / "objects" / f, "rb").read(), f
    g2 = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=32))), encoders=enc, lib=L)
    g2.load_hmsg_graph(str(tmp_path / "graph"))
    assert len(g2.objects) == len(g.objects) and len(g2.rooms) == 1
    if g2.objects:
        # N2, load side: the saved object records parsed by the library straight into a resident index
        from holoagent_amd._lib import HmsgError, NodeIndex
        rid = {r.room_id: i for i, r in enumerate(g2.rooms)}
        ixl = NodeIndex.load_objects(str(tmp_path / "graph" / "objects"), [o.object_id for o in g2.objects],
                                     [rid[o.room_id] for o in g2.objects], lib_=L)
        assert (ixl.N, ixl.D) == (len(g2.objects), 32)
        Tq = g2.get_text_feats_multiple_templates(["chair", "background"])[None]
        a = ixl.query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
        b = g2._node_index().query_objects(Tq, np.zeros(1, np.int32), [[0]], 3)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
        ixl.close()
        with pytest.raises(HmsgError):
            NodeIndex.load_objects(str(tmp_path / "graph" / "objects"), ["no_such_object"], [0], lib_=L)
        fl, rooms2, objs, res = g2.query_hierarchy_protected_icra((None, "office", "chair"), top_k=3)
        emb = np.stack([o.embedding for o in g2.objects])
        T = g2.get_text_feats_multiple_templates(["chair", "background"])
        top, sc = O.query_object(T, 0, emb, 3)
        assert [g2.objects[i].object_id for i in top] == [o.object_id for o in objs]
        np.testing.assert_allclose(res["object_scores"], sc, rtol=0, atol=1e-12)


def test_query_tiled_gemm_small(L):
    """The 128x128-tile float64 MFMA GEMM of the batched query path (ragged edges in M, N and D) against the numpy
    restatement."""
    from holoagent_amd._lib import NodeIndex
    from oracle import hmsg_oracle as O
    rng = np.random.Generator(np.random.PCG64(3))
    N, R, Q, D, k = 203, 5, 71, 27, 4
    emb = rng.standard_normal((N, D)) * 0.1
    room = rng.integers(0, R, size=N).astype(np.int32)
    T = rng.standard_normal((Q, 2, D)).astype(np.float32) * 0.1
    lists = [sorted(rng.choice(R, size=int(rng.integers(1, 4)), replace=False).tolist()) for _ in range(Q)]
    ix = NodeIndex(emb, room, lib_=L)
    idx, rooms, score = ix.query_objects(T, np.zeros(Q, np.int32), lists, k)
    for q in range(Q):
        cand = [o for r in lists[q] for o in np.nonzero(room == r)[0]]
        top, sc = O.query_object(T[q], 0, emb[cand], k)
        assert [cand[t] for t in top] == [int(v) for v in idx[q] if v >= 0]
        np.testing.assert_allclose(score[q][: len(top)], sc, rtol=0, atol=1e-12)
    S = ix.similarity(T[:, 0, :])
    np.testing.assert_allclose(S, np.dot(T[:, 0, :].astype(np.float64), emb.T), rtol=0, atol=1e-12)
    ix.close()


def test_query_exact_score_ties(L):
    """Duplicate embeddings give bit-equal scores.  The negative-prompt path orders them like np.argsort(-score) (the
    earlier candidate first, graph.py:3147-3150).  The plain path's np.argsort(sim)[::-1] (:3133) has no defined order
    for equal keys (numpy's default sort is not stable), so there the scores and the set of nodes per score are compared;
    the library lists tied nodes by candidate position."""
    from holoagent_amd._lib import NodeIndex
    from oracle import hmsg_oracle as O
    rng = np.random.Generator(np.random.PCG64(12))
    D = 12
    base = rng.standard_normal((5, D))
    emb = base[[0, 1, 0, 2, 1, 0, 3, 4, 2]]                       # 9 nodes, rows 0/2/5, 1/4 and 3/8 are duplicates
    room = np.zeros(len(emb), np.int32)
    T = np.stack([base[0] + 0.01 * rng.standard_normal(D), -base[0]]).astype(np.float32)[None]
    ix = NodeIndex(emb, room, lib_=L)
    idx, _, score = ix.query_objects(T, np.zeros(1, np.int32), [[0]], 6, use_negatives=True)
    top, sc = O.query_object(T[0], 0, emb, 6, has_negatives=True)
    assert [int(v) for v in idx[0] if v >= 0] == [int(t) for t in top]
    np.testing.assert_array_equal(score[0][: len(top)], sc)
    idx, _, score = ix.query_objects(T, np.zeros(1, np.int32), [[0]], 9, use_negatives=False)      # (all nodes: no tie
    top, sc = O.query_object(T[0], 0, emb, 9, has_negatives=False)                               #  is cut by k)
    np.testing.assert_array_equal(score[0], sc)
    for v in np.unique(sc):
        assert sorted(int(i) for i in idx[0][score[0] == v]) == sorted(int(t) for t in top[sc == v])
    assert [int(i) for i in idx[0][:3]] == [0, 2, 5]             # tied nodes in candidate order
    ix.close()


_ACCUM_SNIPPET = """
import sys, json, hashlib
sys.path.insert(0, {root!r})
import numpy as np
from tests import golden_io as GI
from tests import parity_common as PC
from holoagent_amd._lib import HmsgLib
L = HmsgLib({lib!r})
z = GI.load("build_hier")
frames = GI.unpack_frames(z)[:6]
cfg = GI.unpack_cfg(z)
cfg["outlier_nb"] = 300
sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300))
PC.check_map(sc, frames, cfg)               # (bit-identical with the oracle's ordered float64 sums)
pts, cols = sc.map_points(colors=True)
print("DIGEST", json.dumps([int(pts.shape[0]), hashlib.sha1(pts.tobytes()).hexdigest()]))
"""


def _accum_routes(lib):
    """The ordered voxel sums (A2, hmsg_map.hip) by their two routes -- the points of a pack staged in LDS and three lanes adding a
    coordinate each (round 6, the default), and every lane of the wave carrying all three sums (the form until round 5,
    HMSG_DEBUG_ACCUM_WAVES=1).  The switch is read once per process: two processes, each compared with the oracle, one cloud."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for over in ({}, {"HMSG_DEBUG_ACCUM_WAVES": "1"}):
        env = dict(os.environ)
        env.pop("HMSG_DEBUG_ACCUM_WAVES", None)
        env.update(over)
        r = subprocess.run([sys.executable, "-c", _ACCUM_SNIPPET.format(root=root, lib=lib)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1][7:]))
    assert out[0] == out[1] and out[0][0] > 100, out


def test_ordered_voxel_sums_two_routes():
    _accum_routes(PC.EMU_PATH)
