"""configs[4] building block: an episode handed over in CHUNKS (hmsg_add_frames appends, hmsg_add_frame_features turns each
chunk's masks into resident bitsets) gives the same scene as one hand-over, bit for bit -- on the kernel simulator here,
scripts/bench_stream_episode.py drives it at 1280x720 on the GPU."""
import os

import numpy as np
import pytest

from tests import parity_common as PC

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


def _run(L, frames, chunk):
    S = PC.stack_frames(frames)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, merge_type=0, outlier_nb_points=20, outlier_radius=0.3))
    n = len(frames)
    for a in range(0, n, chunk):
        b = min(n, a + chunk)
        sc.add_frames(S["rgb"][a:b], S["depth"][a:b], S["pose"][a:b], S["K"])
        sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
    sc.finalize_map()
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    out = dict(map=sc.map_points(), feats=sc.map_feats(counter=True), inst=sc.instances(), pooled=sc.instance_feats())
    sc.close()
    return out


def test_chunked_handover_equals_single_handover():
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=11, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                     n_frames=5, n_masks=8, feat_dim=16, yaw_step_deg=25.0)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    L = HmsgLib(PC.EMU_PATH)
    one, chunks = _run(L, frames, 5), _run(L, frames, 2)
    # a very long episode makes the merge fold compact its point pool and drop its grid arenas now and then
    # (Merger::collect): forced after every frame here, the instances must not change
    os.environ["HMSG_DEBUG_GC_POINTS"] = "1"
    try:
        collected = _run(L, frames, 5)
    finally:
        del os.environ["HMSG_DEBUG_GC_POINTS"]
    assert len(collected["inst"]) == len(one["inst"]) and all(np.array_equal(x, y) for x, y in zip(one["inst"], collected["inst"]))
    assert np.array_equal(one["pooled"], collected["pooled"])
    assert np.array_equal(one["map"], chunks["map"])
    assert np.array_equal(one["feats"][0], chunks["feats"][0]) and np.array_equal(one["feats"][1], chunks["feats"][1])
    assert len(one["inst"]) == len(chunks["inst"]) > 1
    for x, y in zip(one["inst"], chunks["inst"]):
        assert np.array_equal(x, y)
    assert np.array_equal(one["pooled"], chunks["pooled"])
