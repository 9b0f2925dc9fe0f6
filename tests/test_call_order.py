"""C-ABI contract on the kernel simulator: calls out of order or with bad arguments return an HMSG_ERR_* code with a
message (no crash, no silent success), and the smallest episode (one frame) goes through the whole path."""
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


def _frames(n):
    z = GI.load("build_hier")
    return GI.unpack_frames(z)[:n], GI.unpack_cfg(z)


def test_calls_out_of_order_are_refused(L):
    from holoagent_amd._lib import HmsgError, Scene
    frames, cfg = _frames(2)
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=300))
    with pytest.raises(HmsgError, match="geometry"):                       # features for a frame that was never added
        sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    with pytest.raises(HmsgError):                                         # nothing fused yet
        sc.merge_instances()
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    with pytest.raises(HmsgError, match="full"):                           # more frames than cfg.max_frames
        sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    with pytest.raises(HmsgError, match="order"):                          # frames must be handed over in order
        sc.add_frame_features(1, S["masks"][1:], S["f_g"][1:], S["f_masked"][1:], S["f_crop"][1:], S["n_masks"][1:])
    sc.finalize_map()
    with pytest.raises(HmsgError, match="finalise"):
        sc.finalize_map()
    with pytest.raises(HmsgError):                                         # the map is closed
        sc.add_frames(S["rgb"][:0], S["depth"][:0], S["pose"][:0], S["K"])
    with pytest.raises(HmsgError):                                         # pooling needs merged instances
        sc.pool_instances()
    bad = np.array(S["n_masks"]).copy()
    bad[0] = S["masks"].shape[1] + 1
    with pytest.raises(HmsgError, match="n_masks"):
        sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], bad)
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.merge_instances()
    with pytest.raises(HmsgError, match="already"):
        sc.merge_instances()
    sc.close()
    with pytest.raises(HmsgError):                                         # bad configuration: more than 256 masks
        Scene(lib_=L, height=8, width=8, max_frames=1, max_masks=300, feat_dim=4)


def test_single_frame_episode(L):
    frames, cfg = _frames(1)
    cfg["outlier_nb"] = 50
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], outlier_nb_points=50))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    sc.merge_instances()
    sc.pool_instances()
    inst = sc.instances()
    assert len(inst) == sc.num_instances() and sc.instance_feats().shape == (len(inst), cfg["feat_dim"])
    # one frame: every instance is one of the frame's own 3-D masks after the keep-largest DBSCAN
    masks = [m for m in sc.frame_masks3d(0) if len(m)]
    assert 0 < len(inst) <= len(masks)
    sc.close()
