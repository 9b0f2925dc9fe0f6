"""The ROOM level (A9 + N1) from posed RGB-D FRAMES on, with the scene resident on the device -- against a run of the
reference's own create_feature_map (map), segment_floors_manually and segment_hmsg_room on the same frames
(tests/golden/rooms_frames.npz, made by `python -m oracle.refdrive.gen_golden_rooms --frames`; OpenCV / Open3D stand-ins as
for the other fixtures, so N1's operator internals stay "statistical"):

  frames -> hmsg_add_frames / hmsg_finalize_map -> hmsg_segment_floors -> Graph.segment_hmsg_room with NOTHING handed in:
  regions by the device watershed (hmsg_segment_rooms), room clouds by hmsg_room_clouds, camera -> room distances by
  hmsg_points_min_dist_2d, KMeans views, Room and View nodes -- every one of them compared with the reference run, and the
  room clouds also with the cKDTree statement of oracle/rooms_oracle.room_cloud."""
import os

import numpy as np
import pytest

from oracle import rooms_oracle as R
from tests import parity_common as PC

FIX = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rooms_frames.npz")


def _sorted_rows(p):
    p = np.asarray(p, np.float64)
    return p[np.lexsort((p[:, 2], p[:, 1], p[:, 0]))]


def check_rooms_from_frames(L, early=False):
    from holoagent_amd._lib import Scene
    from holoagent_amd.graph import Graph
    z = np.load(FIX)
    F, H, W = z["depth"].shape
    D = z["f_g"].shape[1]
    sc = Scene(lib_=L, height=H, width=W, max_frames=F, max_masks=1, feat_dim=D)
    sc.add_frames(np.ascontiguousarray(z["rgb"]), np.ascontiguousarray(z["depth"]), np.ascontiguousarray(z["pose"]),
                  np.ascontiguousarray(z["K"], np.float64))
    sc.finalize_map()
    assert np.array_equal(_sorted_rows(sc.map_points()), _sorted_rows(z["ref_cloud"]))      # the reference run's map, bit for bit
    g = Graph.from_scene(sc, cfg=dict(main=dict(), models=dict(clip=dict(feat_dim=D)), pipeline=dict(grid_resolution=0.05, skip_frames=1)),
                         lib=L, instances=False)
    g._poses = [np.asarray(p, np.float64) for p in z["pose"]]
    g.set_view_feats(z["f_g"])
    if early:                                   # the same through start_room_level (KMeans on a host thread) + the pick-up
        g.start_room_level()
        box = g._room_level
        box["thread"].join()
        assert box["err"] is None
        g._room_level = None
        for fl, ctx in zip(g.floors, box["ctxs"]):
            g._rooms_finish(fl, ctx)
    else:
        ranges = g.segment_floors_manually(None)
        assert np.array_equal(np.asarray(ranges, np.float64).reshape(-1, 2), z["floor_ranges"])
        for fl in g.floors:
            assert g.segment_hmsg_room(fl) is not None
    assert [f.floor_zero_level for f in g.floors] == list(z["floor_zero"]) and [f.floor_height for f in g.floors] == list(z["floor_height"])
    assert len(g.rooms) == int(z["n_rooms"]) == 2
    for i, room in enumerate(g.rooms):
        fl = g.floors[int(z["room_floor"][i])]
        assert room.room_id == str(z["id_%d" % i]) and room.floor_id == fl.floor_id
        assert np.array_equal(room.vertices, z["vertices_%d" % i]), i                   # device watershed == the reference run's regions
        rp = np.asarray(room.pcd.points)
        assert np.array_equal(_sorted_rows(rp), z["cloud_%d" % i]), i                   # hmsg_room_clouds == the reference run's room cloud
        fp = np.asarray(fl.pcd.points)
        assert np.array_equal(rp, fp[R.room_cloud(fp, room.vertices, fl.floor_zero_level, fl.floor_height)]), i   # ... and the cKDTree statement, in order
        assert list(room.sample_images) == list(z["sample_%d" % i])
        assert list(room.represent_images) == list(z["represent_%d" % i])
        assert np.array_equal(np.asarray(room.embeddings, np.float32).reshape(len(room.represent_images), -1), z["emb_%d" % i])
    assert [v.view_id for v in g.views] == list(z["view_ids"])
    assert [v.room_id for v in g.views] == list(z["view_room"])
    assert [v.img_id for v in g.views] == list(z["view_img"])
    sc.close()


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
@pytest.mark.parametrize("early", [pytest.param(False, marks=pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="a minute on the kernel simulator (HMSG_EMU_SLOW=1); both orders run on the MI355X")), True])
def test_rooms_from_frames_equal_the_reference_run_emu(early):
    from holoagent_amd._lib import HmsgLib
    check_rooms_from_frames(HmsgLib(PC.EMU_PATH), early)


@pytest.mark.gpu
@pytest.mark.parametrize("early", [False, True])
def test_rooms_from_frames_equal_the_reference_run_gpu(early):
    from holoagent_amd._lib import HmsgLib
    check_rooms_from_frames(HmsgLib(), early)
