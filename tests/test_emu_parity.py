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
