"""The CPU oracle (oracle/hmsg_oracle.py) against the golden vectors produced by the reference's own
Python (oracle/refdrive/gen_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import hmsg_oracle as O
from tests import golden_io as GI


def test_fusion_matches_reference():
    z = GI.load("fusion")
    M, H, W = z["shape"]
    masks = np.unpackbits(z["masks"], axis=-1)[..., :W].astype(bool)
    f_p = O.fuse_mask_feats(z["f_g"], z["f_masked"], z["f_crop"], 0.4418)
    np.testing.assert_allclose(f_p, z["ref_f_p"], rtol=0, atol=3e-7)
    f2d = O.per_pixel_feats(masks, f_p).reshape(H, W, -1)
    ref = z["ref_f2d"]
    assert f2d.dtype == np.float16 and ref.dtype == np.float16
    diff = np.abs(f2d.astype(np.float32) - ref.astype(np.float32))
    # fp16 rounding knife edges may flip one ulp on a handful of elements
    assert (diff > 0).mean() < 2e-3
    assert diff.max() <= 2.0 ** -11


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_feats_dbscan_matches_reference(case):
    z = GI.load("feats_dbscan")
    out = np.asarray(O.feats_denoise_dbscan(z["in_" + case], eps=0.01, min_points=100), np.float32)
    np.testing.assert_array_equal(out, z["ref_" + case])


@pytest.mark.parametrize("name", ["build_seq", "build_hier", "build_ragged"])
def test_create_feature_map_matches_reference(name):
    z = GI.load(name)
    frames = GI.unpack_frames(z)
    cfg = GI.unpack_cfg(z)
    res = O.create_feature_map(frames, cfg)
    assert np.array_equal(res["cloud_pts"], z["ref_cloud"])
    ff, rf = res["full_feats"], z["ref_full_feats"]
    assert ff.shape == rf.shape
    d = np.abs(ff - rf)
    assert (d > 1e-6).mean() < 1e-3 and d.max() < 1e-3
    off = z["ref_mask_off"]
    assert len(res["mask_pcds"]) == len(off) - 1
    for i, (p, _c) in enumerate(res["mask_pcds"]):
        assert np.array_equal(p, z["ref_mask_pts"][off[i]:off[i + 1]])
    mf = np.stack([np.asarray(f).reshape(-1) for f in res["mask_feats"]])
    np.testing.assert_allclose(mf, z["ref_mask_feats"], rtol=0, atol=1e-5)


def test_query_matches_reference():
    z = GI.load("query")
    words = [str(w) for w in z["table_words"]]
    table = {w: z["table"][i] for i, w in enumerate(words)}
    obj_emb, obj_room = z["obj_emb"], z["obj_room"]
    room_floor, room_name = z["room_floor"], [str(s) for s in z["room_name"]]
    R = len(room_name)
    k = z["ref_obj_idx"].shape[1]
    floor_rooms = {f: [r for r in range(R) if room_floor[r] == f] for f in (0, 1)}
    order = np.argsort(z["floor_zero"])
    assert list(z["ref_floor_int"]) == [order[0], order[1]]
    for qi, (q, rq, floor_id, nneg) in enumerate(z["qspec"]):
        rooms_list = list(range(R)) if floor_id == -1 else floor_rooms[floor_id]
        name_feats = np.stack([table[room_name[r]] for r in rooms_list])
        rl = O.query_room_label(table["room%d" % rq], name_feats)
        ref_rl = [int(v) for v in z["ref_rooms_label"][qi] if v >= 0]
        assert rl == ref_rl
        # candidate objects: rooms in rl order (indices into rooms_list), objects in room order
        cand = [o for r in rl for o in range(len(obj_room)) if obj_room[o] == rooms_list[r]]
        negs = ["background"] if nneg == 1 else ["background", "wall"]
        T = np.stack([table["thing%d" % q]] + [table[n] for n in negs])
        top, scores = O.query_object(T, 0, obj_emb[cand], k)
        got = [cand[t] for t in top]
        ref = [int(v) for v in z["ref_obj_idx"][qi] if v >= 0]
        assert got == ref
        np.testing.assert_allclose(scores, z["ref_obj_score"][qi][: len(ref)], rtol=0, atol=1e-12)
        # view-embedding room ranking
        voff = z["room_view_off"]
        embs = [z["room_view_emb"][voff[r]:voff[r + 1]] for r in rooms_list]
        keys = [sum(1 for rr in range(r) if room_floor[rr] == room_floor[r]) for r in rooms_list]
        rv = O.query_room_views(table["room%d" % rq], embs, keys, 5)
        ref_rv = [int(v) for v in z["ref_rooms_view"][qi] if v >= 0]
        assert rv == ref_rv
