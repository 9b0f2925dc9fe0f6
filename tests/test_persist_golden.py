"""A11 (persistence): the JSON records holoagent_amd.graph's Floor / Room / Object / View .save() write against
the ones the REFERENCE's own classes wrote for the same small graph (tests/golden/persist.json, made by
oracle/refdrive/gen_golden.py persist): same file names, same keys, same values -- a graph saved by either side
loads on the other.  Also a save -> load round trip through the mirror."""
import json
import os

import numpy as np

from tests import golden_io as GI


def _build():
    from holoagent_amd.graph import Floor, Object, Room, View, _Pcd
    from oracle.refdrive.gen_golden import build_persist_graph, persist_case
    return build_persist_graph(persist_case(), Floor, Room, Object, View, lambda p: _Pcd(p))


def test_saved_records_match_reference(tmp_path):
    ref = json.load(open(os.path.join(GI.GOLDEN, "persist.json")))
    fl, rooms, objects, views = _build()
    for kind, nodes in (("floors", [fl]), ("rooms", rooms), ("objects", objects), ("views", views)):
        d = tmp_path / kind
        d.mkdir()
        for n in nodes:
            n.save(str(d))
        got = {f: json.load(open(d / f)) for f in sorted(os.listdir(d)) if f.endswith(".json")}
        assert sorted(got) == sorted(ref[kind]), kind
        for f in got:
            assert got[f] == ref[kind][f], (kind, f)
        if kind != "views":                                   # every node with a cloud has its .ply next to the record
            assert sorted(p for p in os.listdir(d) if p.endswith(".ply")) == [f[:-5] + ".ply" for f in sorted(got)]


def test_round_trip(tmp_path):
    from holoagent_amd.graph import Object, Room
    fl, rooms, objects, views = _build()
    for n in rooms + objects:
        n.save(str(tmp_path))
    r = Room(rooms[0].room_id, None)
    r.load_new(str(tmp_path))
    assert r.name == rooms[0].name and r.floor_id == rooms[0].floor_id
    np.testing.assert_array_equal(r.vertices, rooms[0].vertices)
    np.testing.assert_array_equal(r.pcd.points, rooms[0].pcd.points)
    o = Object(objects[0].object_id, None)
    o.load_new(str(tmp_path))
    assert o.room_id == objects[0].room_id and o.name == objects[0].name and o.view_ids == objects[0].view_ids
    np.testing.assert_array_equal(o.embedding, np.asarray(objects[0].embedding, np.float64))
    np.testing.assert_array_equal(o.pcd.points, objects[0].pcd.points)


def test_room_type_votes_match_reference():
    """Room.infer_room_type_from_view_embedding against the reference's own (room.py:131-172; tests/golden/roomnames.npz):
    per-view arg-max, majority vote with np.unique's tie order, "unknown room type" for a room without views."""
    from holoagent_amd.graph import Room
    z = GI.load("roomnames")
    types = [str(t) for t in z["types"]]
    off = np.concatenate([[0], np.cumsum(z["counts"])])
    for k, ref in enumerate(z["names"]):
        room = Room("0_%d" % k, "0")
        room.embeddings = [e for e in z["embs"][off[k]:off[k + 1]]]
        assert room.infer_room_type_from_view_embedding(types, z["text"]) == str(ref)


def test_merge_objects_matches_reference():
    """Room.merge_objects (same-name fusion, optional post-pass of build_hier_multimodal_scene_graph) against the
    reference's own room.py:62-129 / object.py:93-103 on ten objects with pairs, a chain, an empty cloud and
    look-alikes of another name (tests/golden/mergeobjects.npz): ids, names, clouds, embeddings, vertices."""
    from holoagent_amd.graph import Object, Room, _Pcd
    from oracle.refdrive.gen_golden import mergeobj_case
    z = GI.load("mergeobjects")
    room = Room("0_3", "0")
    for k, (name, pts, emb) in enumerate(mergeobj_case()):
        o = Object("0_3_%d" % k, "0_3", name=name)
        o.pcd, o.embedding, o.vertices = _Pcd(pts.copy()), emb.copy(), pts[:, [0, 2]].copy()
        room.add_object(o)
    room.merge_objects()
    assert len(room.objects) == int(z["n"])
    assert [o.object_id for o in room.objects] == [str(v) for v in z["ids"]]
    assert [o.name for o in room.objects] == [str(v) for v in z["names"]]
    assert [len(o.pcd.points) for o in room.objects] == z["npts"].tolist()
    np.testing.assert_array_equal(np.concatenate([np.asarray(o.pcd.points).reshape(-1, 3) for o in room.objects]), z["pts"])
    np.testing.assert_allclose(np.stack([np.asarray(o.embedding, np.float64) for o in room.objects]), z["emb"], rtol=0, atol=1e-15)
    for k, o in enumerate(room.objects):
        np.testing.assert_allclose(np.asarray(o.vertices, np.float64), z["vertices_%d" % k], rtol=0, atol=1e-15)


def test_library_prints_floats_like_python_repr():
    """hmsg_save_objects writes JSON numbers itself: its formatter must equal float.__repr__ (what json.dump uses)."""
    import ctypes as C
    import os

    import pytest

    from tests import parity_common as PC
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    from holoagent_amd._lib import HmsgLib
    L = HmsgLib(PC.EMU_PATH)
    rng = np.random.Generator(np.random.PCG64(2))
    vals = np.concatenate([
        rng.integers(0, 2 ** 64, 20000, dtype=np.uint64).view(np.float64),            # every exponent, NaNs included
        rng.standard_normal(20000), rng.standard_normal(5000).astype(np.float32).astype(np.float64),
        10.0 ** rng.integers(-30, 30, 3000) * rng.integers(1, 1000, 3000),
        np.array([0.0, -0.0, 1.0, -1.0, 0.1, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9999999999999998.0, 1e17, 123456789012345680.0,
                  5e-324, 1.7976931348623157e308, np.inf, -np.inf, np.nan, 1e22, 1e23, 2.5, 100.0, 0.001, 1234.5678])])
    vals = np.ascontiguousarray(vals)
    buf = C.create_string_buffer(len(vals) * 34)
    n = L.c.hmsg_test_format_doubles(vals.ctypes.data_as(C.c_void_p), len(vals), buf, len(buf))
    assert n > 0
    got = buf.raw[:n].decode().split("\n")[:-1]
    import json
    want = [json.dumps(float(v)) for v in vals]
    assert len(got) == len(want)
    bad = [(g, w) for g, w in zip(got, want) if g != w]
    assert not bad, bad[:5]


def test_node_records_through_the_c_abi_equal_json_dump(tmp_path):
    """hmsg_write_json / hmsg_write_ply / hmsg_read_json_numbers (include/hmsg.h): floors / rooms / views records written by the
    library are byte for byte what json.dump writes (floor.py:37-52, room.py:309-337, view.py:56-74 key orders), for awkward
    numbers too; the numbers read back are json.load's."""
    import json
    import os
    import numpy as np
    from holoagent_amd._lib import HmsgLib, read_json_numbers, write_json_record, write_ply
    from holoagent_amd.graph import _read_ply
    from tests import parity_common as PC
    path = PC.EMU_PATH if os.path.exists(PC.EMU_PATH) else None
    L = HmsgLib(path)                       # (host-side code: the product library and the simulator build share it)
    rng = np.random.default_rng(3)
    awkward = np.array([0.0, -0.0, 1e-7, 123456789012345678.0, 1e16, 9999999999999998.0, 1.5e-5, 0.1, -2.5e+22, 5e-324, 1 / 3])
    emb32 = [rng.standard_normal(9).astype(np.float32) for _ in range(4)]
    verts = np.concatenate([rng.standard_normal((6, 2)) * 10, awkward[:10].reshape(5, 2)])
    meta = dict(room_id="1_3", name="kitchen \\u00e9", floor_id="1", objects=["1_3_0", "1_3_10"], views=["1_3_7"], vertices=verts.tolist(),
                room_height=float(awkward[6]), room_zero_level=-0.25, embeddings=[e.tolist() for e in emb32], represent_images=[12, 0, 7],
                sample_images=[], clip_embeddings=[e.tolist() for e in emb32[:2]])
    json.dump(meta, open(tmp_path / "py.json", "w"))
    write_json_record(tmp_path / "c.json", [("room_id", "1_3"), ("name", "kitchen \\u00e9"), ("floor_id", "1"), ("objects", ["1_3_0", "1_3_10"]),
                                            ("views", ["1_3_7"]), ("vertices", verts), ("room_height", float(awkward[6])),
                                            ("room_zero_level", -0.25), ("embeddings", np.stack(emb32)), ("represent_images", np.array([12, 0, 7])),
                                            ("sample_images", []), ("clip_embeddings", np.stack(emb32[:2]))], L)
    assert open(tmp_path / "py.json", "rb").read() == open(tmp_path / "c.json", "rb").read()
    assert np.array_equal(read_json_numbers(tmp_path / "c.json", "vertices", L), np.asarray(json.load(open(tmp_path / "py.json"))["vertices"]).ravel())
    assert np.array_equal(read_json_numbers(tmp_path / "c.json", "embeddings", L), np.stack(emb32).astype(np.float64).ravel())
    assert list(read_json_numbers(tmp_path / "c.json", "represent_images", L)) == [12.0, 0.0, 7.0]
    assert len(read_json_numbers(tmp_path / "c.json", "sample_images", L)) == 0
    pts = rng.standard_normal((17, 3))
    write_ply(tmp_path / "c.ply", pts, L)
    assert np.array_equal(_read_ply(str(tmp_path / "c.ply")), pts)
