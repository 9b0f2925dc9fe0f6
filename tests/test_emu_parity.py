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
    frames = GI.unpack_frames(z)[:10]
    cfg = GI.unpack_cfg(z)
    cfg["outlier_nb"] = 300            # 10 low-res frames: keep a useful part of the cloud
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    assert 0 < ref_pts.shape[0] < sc.map_size_unfiltered()
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    sc.close()


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
    sc.close()


def test_query_golden(L):
    PC.check_query_golden(L)
