"""A8 (floors): holoagent_amd.graph.Graph.segment_floors_manually against what the REFERENCE's own
Graph.segment_floors_manually (graph.py:624-787) produced on the same clouds (tests/golden/floors.npz, made by
oracle/refdrive/gen_golden.py floors).  Height ranges / zero levels / heights to 1e-9 (the voxel re-sampling
inside differs from a sequential float64 sum by <= 1e-13), floor point counts exact."""
import os

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC


def _check(L):
    from holoagent_amd.graph import Graph, _Pcd
    z = GI.load("floors")
    zb = GI.load("build_hier")
    for name in [str(c) for c in z["cases"]]:
        pts = np.asarray(zb["ref_cloud"], np.float64) if name == "fixture_scene" else z[name + "_pts"]
        g = Graph(dict(main=dict(), models=dict(clip=dict(feat_dim=8))), lib=L)
        g.full_pcd = _Pcd(pts)
        ranges = np.array(g.segment_floors_manually(None), dtype=np.float64).reshape(-1, 2)
        ref = z[name + "_ranges"]
        assert ranges.shape == ref.shape, (name, ranges, ref)
        np.testing.assert_allclose(ranges, ref, rtol=0, atol=1e-9, err_msg=name)
        np.testing.assert_allclose([f.floor_zero_level for f in g.floors], z[name + "_zero"], rtol=0, atol=1e-9)
        np.testing.assert_allclose([f.floor_height for f in g.floors], z[name + "_height"], rtol=0, atol=1e-9)
        assert [len(f.pcd.points) for f in g.floors] == z[name + "_npts"].tolist()
        np.testing.assert_allclose(np.stack([f.vertices for f in g.floors]), z[name + "_vertices"], rtol=0, atol=1e-9)


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_floors_match_reference_simulator():
    from holoagent_amd._lib import HmsgLib
    _check(HmsgLib(PC.EMU_PATH))


@pytest.mark.gpu
def test_floors_match_reference_gpu():
    from holoagent_amd._lib import HmsgLib
    _check(HmsgLib())


def _check_vds(L):
    """hmsg_voxel_down_sample against the oracle's Open3D restatement on a cloud whose grid spans many scan tiles:
    same voxel set and order, centroids to 1e-12 (fixed-point sums), single-point voxels bit for bit."""
    from holoagent_amd._lib import Scene
    from oracle import hmsg_oracle as O
    rng = np.random.Generator(np.random.PCG64(8))
    pts = np.concatenate([rng.uniform(-4.0, 6.0, size=(60000, 3)),                 # sparse: mostly one point per voxel
                          rng.normal(0.0, 0.15, size=(40000, 3)) + [1.0, 0.5, -2.0]])  # dense blob: many per voxel
    sc = Scene(lib_=L, height=8, width=8, max_frames=1, max_masks=1, feat_dim=8)
    got = sc.voxel_down_sample(pts, 0.05)
    sc.close()
    ref, _, _, _ = O.o3d_voxel_down_sample(pts, None, 0.05)
    assert got.shape == ref.shape
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-12)
    idx, _ = O.o3d_voxel_keys(pts, 0.05)
    dims = idx.max(axis=0) + 1
    _, counts = np.unique((idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2], return_counts=True)   # canonical order
    single = counts == 1
    assert single.sum() > 1000 and (~single).sum() > 1000 and np.array_equal(got[single], ref[single])


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_voxel_down_sample_simulator():
    from holoagent_amd._lib import HmsgLib
    _check_vds(HmsgLib(PC.EMU_PATH))


@pytest.mark.gpu
def test_voxel_down_sample_gpu():
    from holoagent_amd._lib import HmsgLib
    _check_vds(HmsgLib())
