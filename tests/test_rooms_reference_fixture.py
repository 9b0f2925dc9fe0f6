"""Room level (A9 + N1) against a run of the REFERENCE's own Graph.segment_hmsg_room (graph.py:920-1189) and
distance_transform (graph_utils.py:391-487): tests/golden/rooms.npz, made by oracle/refdrive/gen_golden_rooms.py with a
stand-in cv2 restating the documented semantics of the calls those functions make (OpenCV is not in this image).  What
it pins is everything the reference does around them -- slab slices, histogram orientation, border, morphology order, seed
filter, background marker, grid -> point mapping, room clouds, camera -> room assignment, Room fields and View ids:

  * oracle/rooms_oracle.segment_rooms (the function the HIP path is compared with pixel for pixel in
    test_rooms_segmentation.py) gives the reference run's room masks and region points, bit for bit;
  * the mirror's Graph.segment_hmsg_room, handed those regions, gives the reference run's room clouds, sample /
    representative images, embeddings and View nodes.
"""
import os

import numpy as np
import pytest

from oracle import rooms_oracle as R

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = os.path.join(HERE, "golden", "rooms.npz")


@pytest.fixture(scope="module")
def fx():
    return np.load(FIX)


def _regions(markers, n, xz_min, res):
    out = []
    for i in range(n):
        y_cells, x_cells = np.where(markers == i + 1)
        out.append(np.column_stack(((x_cells - 10.5) * res + xz_min[0], (y_cells - 10.5) * res + xz_min[1])))
    return out


def test_oracle_regions_equal_the_reference_run(fx):
    markers, n, xz_min = R.segment_rooms(fx["cloud"], float(fx["zero_level"]), float(fx["height"]), 0.05)
    assert n == int(fx["n_rooms"]) == 2
    shape = tuple(fx["mask_shape"])
    assert markers.shape == shape
    masks = np.unpackbits(fx["room_masks"], axis=-1)[..., :shape[1]].astype(bool)
    for i, reg in enumerate(_regions(markers, n, xz_min, 0.05)):
        assert np.array_equal(masks[i], markers == i + 1), i
        assert np.array_equal(reg, fx["vertices_%d" % i]), i        # map_grid_to_point_cloud, same doubles
    # the two rooms are the two halves of the storey, split at the dividing wall (x = 3.6)
    cx = sorted(float(fx["vertices_%d" % i][:, 0].mean()) for i in range(2))
    assert cx[0] < 3.0 < 3.6 < cx[1]


def test_mirror_room_nodes_equal_the_reference_run(fx):
    from holoagent_amd.graph import Floor, Graph, _Pcd
    from tests import parity_common as PC
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    from holoagent_amd._lib import HmsgLib
    g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=int(fx["feats"].shape[1]))), pipeline=dict(grid_resolution=0.05)),
              lib=HmsgLib(PC.EMU_PATH))
    fl = Floor("0", name="floor_0")
    fl.pcd = _Pcd(fx["cloud"])
    fl.floor_zero_level, fl.floor_height = float(fx["zero_level"]), float(fx["height"])
    g.floors.append(fl)
    g._poses = list(fx["poses"])
    g._view_feats = [f.reshape(1, -1) for f in fx["feats"]]
    n = int(fx["n_rooms"])
    g.segment_hmsg_room(fl, None, room_2d_points=[fx["vertices_%d" % i] for i in range(n)])
    assert len(g.rooms) == n
    for i, room in enumerate(g.rooms):
        assert room.room_id == str(fx["id_%d" % i])
        assert np.array_equal(np.asarray(room.pcd.points), fx["cloud"][fx["cloud_idx_%d" % i]]), i
        assert list(room.sample_images) == list(fx["sample_%d" % i])
        assert list(room.represent_images) == list(fx["represent_%d" % i])
        assert np.array_equal(np.asarray(room.embeddings, np.float32).reshape(len(room.represent_images), -1), fx["emb_%d" % i])
    assert [v.view_id for v in g.views] == list(fx["view_ids"])
    assert [v.room_id for v in g.views] == list(fx["view_room"])
    assert [v.img_id for v in g.views] == list(fx["view_img"])
