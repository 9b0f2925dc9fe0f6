"""N2: Room.merge_objects behind the C ABI (include/hmsg.h: hmsg_merge_room_objects; holoagent_amd/csrc/hmsg_objmerge.hip) --
the same-name overlap tests on the device, the reference's dictionary chaining and list(set()) order restated in C++.
Against (a) tests/golden/mergeobjects.npz, a run of the reference's own room.py:62-129 / object.py:93-103, and (b) the Python
mirror (numpy / cKDTree overlap tests, Python's own dict and set) on randomised rooms, including rooms whose groups grow past the
sizes at which CPython's set re-hashes."""
import os

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC


def _golden_room(scene):
    from holoagent_amd.graph import Object, Room, _Pcd
    from oracle.refdrive.gen_golden import mergeobj_case
    room = Room("0_3", "0")
    for k, (name, pts, emb) in enumerate(mergeobj_case()):
        o = Object("0_3_%d" % k, "0_3", name=name)
        o.pcd, o.embedding, o.vertices = _Pcd(pts.copy()), emb.copy(), pts[:, [0, 2]].copy()
        room.add_object(o)
    room.merge_objects(scene=scene)
    return room


def _check_golden(room):
    z = GI.load("mergeobjects")
    assert len(room.objects) == int(z["n"])
    assert [o.object_id for o in room.objects] == [str(v) for v in z["ids"]]
    assert [o.name for o in room.objects] == [str(v) for v in z["names"]]
    assert [len(o.pcd.points) for o in room.objects] == z["npts"].tolist()
    np.testing.assert_array_equal(np.concatenate([np.asarray(o.pcd.points).reshape(-1, 3) for o in room.objects]), z["pts"])
    np.testing.assert_allclose(np.stack([np.asarray(o.embedding, np.float64) for o in room.objects]), z["emb"], rtol=0, atol=1e-15)
    for k, o in enumerate(room.objects):
        np.testing.assert_allclose(np.asarray(o.vertices, np.float64), z["vertices_%d" % k], rtol=0, atol=1e-15)


def _random_rooms(seed, n_rooms):
    rng = np.random.Generator(np.random.PCG64(seed))
    for trial in range(n_rooms):
        n = int(rng.integers(2, 26))
        names = ["n%d" % v for v in rng.integers(0, 1 + trial % 3, n)]
        clouds = []
        for i in range(n):
            kind = trial % 4
            if kind == 0 and i and rng.random() < 0.6:          # copies of an earlier cloud: every such pair overlaps fully
                clouds.append(clouds[int(rng.integers(0, i))].copy())
                continue
            m = int(rng.integers(0 if rng.random() < 0.08 else 3, 400))
            centre = rng.uniform(-0.6, 0.6, 3) * (0.3 if kind == 1 else 1.0)
            clouds.append(centre + rng.uniform(-0.15, 0.15, (m, 3)))
        yield names, clouds


def _check_random(scene, seed=11, n_rooms=24):
    from holoagent_amd.graph import Object, Room, _Pcd
    merged_rooms = 0
    for names, clouds in _random_rooms(seed, n_rooms):
        room = Room("0_0", "0")
        for k, (nm, pts) in enumerate(zip(names, clouds)):
            o = Object("0_0_%d" % k, "0_0", name=nm)
            o.pcd = _Pcd(np.asarray(pts, np.float64).reshape(-1, 3))
            room.add_object(o)
        want = room.merge_groups()
        got = scene.merge_room_objects(clouds, names)
        assert got == want, (names, want, got)
        merged_rooms += any(len(g) > 1 for g in want)
    assert merged_rooms >= n_rooms // 3            # (the sweep does merge: chains, re-hashed sets and all)


def _emu_scene():
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    from holoagent_amd._lib import HmsgLib, Scene
    return Scene(lib_=HmsgLib(PC.EMU_PATH), height=8, width=8, max_frames=1, max_masks=1, feat_dim=8)


def test_merge_room_objects_matches_the_reference_run():
    sc = _emu_scene()
    _check_golden(_golden_room(sc))
    sc.close()


def test_merge_room_objects_equals_the_mirror_on_random_rooms():
    sc = _emu_scene()
    _check_random(sc)
    sc.close()


def test_faiss_blas_form_is_honoured():
    """overlap_distance_form = HMSG_OVERLAP_FAISS_BLAS: clouds of 20 or more points are looked up in faiss's BLAS form (as in the merge
    fold); the decisions on the fixture's well separated clouds are the same."""
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    from holoagent_amd._lib import HmsgLib, Scene
    sc = Scene(lib_=HmsgLib(PC.EMU_PATH), height=8, width=8, max_frames=1, max_masks=1, feat_dim=8, overlap_distance_form=1)
    _check_golden(_golden_room(sc))
    sc.close()


@pytest.mark.gpu
def test_merge_room_objects_matches_the_reference_run_gpu():
    from holoagent_amd._lib import HmsgLib, Scene
    sc = Scene(lib_=HmsgLib(), height=8, width=8, max_frames=1, max_masks=1, feat_dim=8)
    _check_golden(_golden_room(sc))
    _check_random(sc, seed=5, n_rooms=40)
    sc.close()
