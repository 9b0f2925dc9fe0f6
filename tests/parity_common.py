"""Shared body of the parity tests: drive the C ABI (include/hmsg.h) stage by stage and compare with the
CPU oracle on the same seeded inputs.  Used with the real gfx950 library by the `-m gpu` tests and with
the kernel-simulator build (tests/emu) by the CPU-side development tests."""
import os

import numpy as np
from scipy.spatial import cKDTree

from oracle import hmsg_oracle as O

# HMSG_EMU_PATH: another build of the simulator (scripts/emu_sanitize.sh points it at an AddressSanitizer build)
EMU_PATH = os.environ.get("HMSG_EMU_PATH") or os.path.join(os.path.dirname(__file__), "emu", "libhmsg_emu.so")


def _pad(a, rows):
    out = np.zeros((rows,) + a.shape[1:], a.dtype)
    out[: a.shape[0]] = a
    return out


def stack_frames(frames):
    """Hand-over layout of the C ABI: frames may hold different numbers of masks (SAM does); rows are padded to the
    largest count and `n_masks` says how many are real."""
    M = max(max(f["masks"].shape[0] for f in frames), 1)
    D = frames[0]["f_g"].reshape(-1).shape[0]
    return dict(
        rgb=np.ascontiguousarray(np.stack([f["rgb"] for f in frames])),
        depth=np.ascontiguousarray(np.stack([f["depth"] for f in frames])),
        pose=np.ascontiguousarray(np.stack([f["pose"] for f in frames])),
        K=np.ascontiguousarray(frames[0]["K"], dtype=np.float64),
        masks=np.ascontiguousarray(np.stack([_pad(f["masks"].astype(np.uint8), M) for f in frames])),
        f_g=np.ascontiguousarray(np.stack([f["f_g"].reshape(-1) for f in frames]).astype(np.float32)),
        f_masked=np.ascontiguousarray(np.stack([_pad(np.asarray(f["f_masked"], np.float32).reshape(-1, D), M) for f in frames])),
        f_crop=np.ascontiguousarray(np.stack([_pad(np.asarray(f["f_crop"], np.float32).reshape(-1, D), M) for f in frames])),
        n_masks=np.array([f["masks"].shape[0] for f in frames], np.int32))


def make_scene(L, frames, cfg_over):
    from holoagent_amd._lib import Scene
    H, W = frames[0]["depth"].shape
    M = max(f["masks"].shape[0] for f in frames)
    over = dict(height=H, width=W, max_frames=len(frames), max_masks=max(M, 1))
    over.update(cfg_over)
    return Scene(lib_=L, **over)


def check_map(sc, frames, cfg, oracle_cloud=None):
    """A1 + A2: identical voxel set / order, centroids BIT-IDENTICAL (ordered float64 sums), colours to 1e-9."""
    S = stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    if oracle_cloud is None:
        oracle_cloud = O.build_global_cloud(frames, cfg["voxel_size"], cfg.get("outlier_nb", 1000),
                                            cfg.get("outlier_radius", 1.0))
    ref_pts, ref_cols, info = oracle_cloud
    assert sc.map_size_unfiltered() == info["n_voxels"]
    assert sc.map_size() == ref_pts.shape[0]
    pts, cols = sc.map_points(colors=True)
    assert np.array_equal(pts, ref_pts), float(np.abs(pts - ref_pts).max())
    np.testing.assert_allclose(cols, ref_cols, rtol=0, atol=1e-9)
    return S, ref_pts, ref_cols


def check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks=True):
    """A3 + A4 + A5: F_p to 3e-7, NN indices identical, counters identical, feature map within 1e-6 (fp16 knife edges
    allowed on a <1e-3 fraction, bounded by one fp16 ulp), 3-D masks BIT-IDENTICAL."""
    O.NN_TIE = "scipy"    # the reference's own behaviour, bit-equal ties included (the library replays cKDTree for those)
    return _check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks)


def _check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks):
    D = cfg["feat_dim"]
    n = len(frames)
    half = n // 2
    for a, b in ((0, half), (half, n)):
        if b > a:
            sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
    sc.fuse_frames()
    tree = cKDTree(ref_pts)
    V = ref_pts.shape[0]
    counter = np.zeros((V, 1), np.float32)
    sums = np.zeros((V, D), np.float32)
    for i, fr in enumerate(frames):
        assert sc.frame_num_masks(i) == fr["masks"].shape[0]
        f_p = O.fuse_mask_feats(fr["f_g"], fr["f_masked"], fr["f_crop"], cfg["clip_masked_weight"])
        np.testing.assert_allclose(sc.frame_fp(i), f_p, rtol=0, atol=3e-7)
        p, _, valid = O.create_pcd(fr["rgb"], fr["depth"], fr["pose"], fr["K"])
        dist, idx = O.nn_query(tree, p)
        got = sc.frame_nn(i)
        assert (got[~valid] == -1).all()
        g = got[valid]
        assert np.array_equal(g, idx), "NN index mismatch in frame %d: %d pixels" % (i, int((g != idx).sum()))
        f2d = O.per_pixel_feats(fr["masks"], f_p) if fr["masks"].shape[0] else np.zeros((valid.size, D), np.float16)
        O.fuse_frame_into_map(sums, counter, f2d, fr["depth"], g.astype(np.int64))
        if check_masks:
            ref_masks = O.create_3d_masks(fr["masks"], fr["depth"], ref_pts, ref_cols, tree, fr["pose"], fr["K"],
                                          cfg["voxel_size"], cfg["max_mask_distance"])
            got_masks = sc.frame_masks3d(i)
            assert len(got_masks) == len(ref_masks)
            for m, (rp, _rc) in enumerate(ref_masks):
                assert got_masks[m].shape == rp.shape, (i, m, got_masks[m].shape, rp.shape)
                assert np.array_equal(got_masks[m], rp), (i, m, float(np.abs(got_masks[m] - rp).max()))
    feats, cnt = sc.map_feats(counter=True)
    np.testing.assert_array_equal(cnt, counter[:, 0])
    c = counter.copy()
    c[c == 0] = 1e-5
    ref_feats = sums / c
    d = np.abs(feats - ref_feats)
    assert (d > 1e-6).mean() < 1e-3, (d > 1e-6).mean()
    assert d.max() <= 2.0 ** -10
    return ref_feats, 0


def check_merge_pool(sc, frames, cfg, ref_pts, ref_feats):
    """A6 + A7, stage-wise: the oracle merges the library's own 3-D masks (already checked bit for bit against the
    oracle's) and pools the library's own instances; instance point sets must be BIT-IDENTICAL (count, order,
    coordinates), pooled features within 1e-5 (the north-star tolerance)."""
    O.NN_TIE = "scipy"
    try:
        frames_pcd = []
        for i in range(len(frames)):
            frames_pcd.append([(p, np.zeros_like(p)) for p in sc.frame_masks3d(i)])
        vs = cfg["voxel_size"]
        if cfg.get("merge_type", "sequential") == "hierarchical":
            ref = O.hierarchical_merge(frames_pcd, cfg["init_overlap_thresh"], cfg["overlap_thresh_factor"], vs,
                                       cfg["iou_thresh"])
        else:
            ref = O.seq_merge(frames_pcd, cfg["init_overlap_thresh"], vs, cfg["iou_thresh"])
        ref = [m for m in ref if m[0].shape[0] >= 10]
        sc.merge_instances()
        got = sc.instances()
        assert len(got) == len(ref), (len(got), len(ref))
        for k, (g, (r, _)) in enumerate(zip(got, ref)):
            assert g.shape == r.shape, (k, g.shape, r.shape)
            assert np.array_equal(g, r), (k, float(np.abs(g - r).max()))
        sc.pool_instances()
        feats = sc.instance_feats()
        tree = cKDTree(ref_pts)
        map_feats = sc.map_feats()
        ref_pool = O.pool_instances([(g, np.zeros_like(g)) for g in got], ref_pts, tree, map_feats, vs, cfg["feat_dim"])
        ref_pool = np.stack([np.asarray(f, np.float32).reshape(-1) for f in ref_pool]) if ref_pool else np.zeros((0, cfg["feat_dim"]))
        assert feats.shape == ref_pool.shape
        np.testing.assert_allclose(feats, ref_pool, rtol=0, atol=1e-5)
        return got, feats
    finally:
        O.NN_TIE = "scipy"


def check_against_reference_run(name, z, sc, got, feats):
    """The library's end result against what the REFERENCE's own create_feature_map produced (tests/golden):
    identical map, identical instances (bit for bit), voxel features within fp16 knife edges, and pooled features
    within 1e-5 for EVERY instance -- including the ones that hinge on a bit-equal nearest-neighbour tie (an instance
    point that is the exact float64 midpoint of two map voxels; the fixture flags them), which the library answers by
    replaying scipy's cKDTree traversal (holoagent_amd/csrc/hmsg_ckdtree.h)."""
    off = z["ref_mask_off"]
    n_ref = len(off) - 1
    assert len(got) == n_ref
    for k in range(n_ref):
        assert np.array_equal(got[k], z["ref_mask_pts"][off[k]:off[k + 1]]), k
    mf = sc.map_feats()
    d = np.abs(mf - z["ref_full_feats"])
    frac_rows = float((d.max(axis=1) > 1e-6).mean())
    hinge = z["ref_tie_sensitive"]
    err = np.abs(feats - z["ref_mask_feats"]).max(axis=1)
    ok = err <= 1e-5
    print("%s: map rows off the reference run %.4f; instances %d, within 1e-5 of the reference run %d (%.1f %%), "
          "hinging on a bit-equal NN tie %d (tie queries answered by the cKDTree replay: %d), max err %.3g"
          % (name, frac_rows, n_ref, int(ok.sum()), 100.0 * ok.mean(), int(hinge.sum()), sc.num_tie_queries(), float(err.max())))
    # (an element on an fp16 rounding boundary of the per-pixel feature can differ by ONE fp16 ulp, 2^-10 at most: the row
    #  norm is a wave reduction here and a vectorised torch reduction in the reference)
    assert (d > 1e-6).mean() < 2e-3 and d.max() <= 2.0 ** -10, (float((d > 1e-6).mean()), float(d.max()))
    assert err.max() <= 1e-5, (int((~ok).sum()), float(err.max()))
    assert sc.num_tie_queries() > 0
    return ok


def check_query_golden(L):
    """A12 against the fixture produced by the reference's own query_hmsg_object (tests/golden/query.npz):
    retrieval indices bit-identical, scores to 1e-12."""
    from holoagent_amd._lib import NodeIndex
    from tests import golden_io as GI
    z = GI.load("query")
    words = [str(w) for w in z["table_words"]]
    table = {w: z["table"][i] for i, w in enumerate(words)}
    obj_emb, obj_room = z["obj_emb"], z["obj_room"]
    room_floor = z["room_floor"]
    R = len(room_floor)
    k = z["ref_obj_idx"].shape[1]
    ix = NodeIndex(obj_emb, obj_room, lib_=L)
    floor_rooms = {f: [r for r in range(R) if room_floor[r] == f] for f in (0, 1)}
    # the fixture's queries, batched per number of negatives
    for nneg in (1, 2):
        qs = [i for i, s in enumerate(z["qspec"]) if s[3] == nneg]
        T, lists = [], []
        for qi in qs:
            q, rq, floor_id, _ = z["qspec"][qi]
            rooms_list = list(range(R)) if floor_id == -1 else floor_rooms[floor_id]
            rl = [int(v) for v in z["ref_rooms_label"][qi] if v >= 0]
            lists.append([rooms_list[r] for r in rl])
            negs = ["background"] if nneg == 1 else ["background", "wall"]
            T.append(np.stack([table["thing%d" % q]] + [table[n] for n in negs]))
        idx, room, score = ix.query_objects(np.stack(T), np.zeros(len(qs), np.int32), lists, k)
        for j, qi in enumerate(qs):
            ref = [int(v) for v in z["ref_obj_idx"][qi] if v >= 0]
            got = [int(v) for v in idx[j] if v >= 0]
            assert got == ref, (qi, got, ref)
            np.testing.assert_allclose(score[j][: len(ref)], z["ref_obj_score"][qi][: len(ref)], rtol=0, atol=1e-12)
            floor_id = z["qspec"][qi][2]
            rooms_list = list(range(R)) if floor_id == -1 else floor_rooms[floor_id]
            ref_room = [rooms_list[int(v)] for v in z["ref_obj_room"][qi] if v >= 0]
            assert [int(v) for v in room[j][: len(ref)]] == ref_room
    # query that IS one of the negative labels (graph.py:3082-3092): categories = negatives, qid = its index
    T = np.stack([table["background"], table["wall"]])[None]
    idx, room, score = ix.query_objects(T, np.array([1], np.int32), [list(range(R))], k)
    assert [int(v) for v in idx[0] if v >= 0] == [int(v) for v in z["ref_neg_idx"]]
    np.testing.assert_allclose(score[0][: len(z["ref_neg_score"])], z["ref_neg_score"], rtol=0, atol=1e-12)
    # the whole coarse-to-fine query on the device (hmsg_query_hier): floors -> rooms (label mode: the reference's own
    # query_hmsg_room results; view mode likewise) -> objects, against the same reference-made fixture
    room_name_emb = np.stack([table[str(n)] for n in z["room_name"]]).astype(np.float64)
    voff = z["room_view_off"]
    views = [z["room_view_emb"][voff[r]:voff[r + 1]] for r in range(R)]
    keys, cnt = [], {0: 0, 1: 0}
    for r in range(R):                                       # room_id = "<floor>_<position on the floor>"
        keys.append(cnt[int(room_floor[r])])
        cnt[int(room_floor[r])] += 1
    ix.set_hierarchy([floor_rooms[0], floor_rooms[1]], room_name_emb, views, keys)
    for nneg in (1, 2):
        qs = [i for i, s in enumerate(z["qspec"]) if s[3] == nneg]
        negs = ["background"] if nneg == 1 else ["background", "wall"]
        T = np.stack([np.stack([table["thing%d" % z["qspec"][qi][0]]] + [table[n] for n in negs]) for qi in qs])
        Tr = np.stack([table["room%d" % z["qspec"][qi][1]] for qi in qs])
        fl = np.array([z["qspec"][qi][2] for qi in qs], np.int32)
        rooms_sel, idx, room, score = ix.query_hier(T, np.zeros(len(qs), np.int32), Tr, fl, np.ones(len(qs), np.int32), k)
        for j, qi in enumerate(qs):
            assert rooms_sel[j] == [int(v) for v in z["ref_rooms_label"][qi] if v >= 0], (qi, rooms_sel[j])
            ref = [int(v) for v in z["ref_obj_idx"][qi] if v >= 0]
            assert [int(v) for v in idx[j] if v >= 0] == ref, (qi, idx[j], ref)
            np.testing.assert_allclose(score[j][: len(ref)], z["ref_obj_score"][qi][: len(ref)], rtol=0, atol=1e-12)
            rooms_list = list(range(R)) if fl[j] == -1 else floor_rooms[int(fl[j])]
            assert [int(v) for v in room[j][: len(ref)]] == [rooms_list[int(v)] for v in z["ref_obj_room"][qi] if v >= 0]
        # the view-embedding branch of query_hmsg_room (top 5 room keys by their best view)
        # (floor -1 on the two-storey graph: rooms "0_2" and "1_2" collapse into one key that keeps the place of its best
        #  occurrence, graph.py:3259-3264 -- the fixture holds the reference's de-duplicated lists)
        ok = list(range(len(qs)))
        rooms_v, _, _, _ = ix.query_hier(T[ok], np.zeros(len(ok), np.int32), Tr[ok], fl[ok], np.full(len(ok), 2, np.int32), k)
        for jj, j in enumerate(ok):
            assert rooms_v[jj] == [int(v) for v in z["ref_rooms_view"][qs[j]] if v >= 0][:5], (qs[j], rooms_v[jj])
    # plain similarity (query_floor / query_hmsg_room GEMV)
    S = ix.similarity(np.stack([table["thing3"], table["room1"]]))
    np.testing.assert_allclose(S, np.dot(np.stack([table["thing3"], table["room1"]]), obj_emb.T), rtol=0, atol=1e-12)
    ix.close()
