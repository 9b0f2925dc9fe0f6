"""The incremental merge fold (holoagent_amd/csrc/hmsg_fold.inl) against the batch fold, bit for bit.

seq_merge (graph_utils.py:1015-1038) is a fold over the frames; the library has two exact implementations of a step
(merge_3d_masks, graph_utils.py:918-956): the batch one re-clusters every touched cloud in full, the incremental one
only looks at the new points and what is within eps of them, on a persistent index.  hmsg_merge_instances starts with
the batch fold and switches when the batches grow (HMSG_FOLD_SWITCH); whatever the switch point, the instances have to
be the same clouds, point for point, in the same order.

CPU: the kernel simulator on a small scene.  GPU: configs[1]-shaped scenes (640x480, 32 masks, device-rendered), up to
the full 1000 frames.
"""
import hashlib
import os

import numpy as np
import pytest

from tests import parity_common as PC

FOLD_ENV = ("HMSG_FOLD_LEGACY", "HMSG_FOLD_INCREMENTAL", "HMSG_FOLD_SWITCH", "HMSG_DEBUG_NOANCHOR")


def _set_mode(mode):
    for k in FOLD_ENV:
        os.environ.pop(k, None)
    if mode == "batch":
        os.environ["HMSG_FOLD_LEGACY"] = "1"
    elif mode == "incremental":
        os.environ["HMSG_FOLD_INCREMENTAL"] = "1"
    elif mode == "incremental-noanchor":
        os.environ["HMSG_FOLD_INCREMENTAL"] = "1"
        os.environ["HMSG_DEBUG_NOANCHOR"] = "1"
    else:
        os.environ["HMSG_FOLD_SWITCH"] = str(mode)


def _merge_host_frames(L, frames, mode, feat_dim):
    _set_mode(mode)
    try:
        sc = PC.make_scene(L, frames, dict(feat_dim=feat_dim, outlier_nb_points=200, feat_dbscan_min=20))
        S = PC.stack_frames(frames)
        sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
        sc.finalize_map()
        sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"])
        sc.fuse_frames()
        sc.merge_instances()
        inst = sc.instances()
        sc.close()
        return inst
    finally:
        _set_mode("default-cleanup")
        os.environ.pop("HMSG_FOLD_SWITCH", None)


def _same(a, b):
    return len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_incremental_fold_equals_batch_fold_on_the_simulator():
    """8 low-resolution frames of one room: the batch fold, the incremental fold from the first step (with and without
    anchors) and a switch in the middle give the same instances."""
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib(PC.EMU_PATH)
    spec = SceneSpec(seed=3, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=80, height=60,
                     n_frames=8, n_masks=6, feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    ref = _merge_host_frames(L, frames, "batch", 16)
    assert len(ref) >= 5 and sum(len(x) for x in ref) > 2000
    for mode in ("incremental", 4000):
        got = _merge_host_frames(L, frames, mode, 16)
        assert _same(ref, got), (mode, [len(x) for x in ref], [len(x) for x in got])


def _merge_device_scene(L, spec, inp, mode):
    from holoagent_amd._lib import Scene
    _set_mode(mode)
    try:
        sc = Scene(lib_=L, height=spec.height, width=spec.width, max_frames=spec.n_frames, max_masks=spec.n_masks, feat_dim=spec.feat_dim)
        sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
        sc.finalize_map()
        sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"])
        sc.fuse_frames()
        sc.merge_instances()
        inst = sc.instances()
        sc.close()
    finally:
        for k in FOLD_ENV:
            os.environ.pop(k, None)
    h = hashlib.sha1()
    for a in inst:
        h.update(np.ascontiguousarray(a).tobytes())
    return len(inst), sum(len(a) for a in inst), h.hexdigest()


@pytest.mark.gpu
@pytest.mark.parametrize("n_frames", [120, 1000])
def test_incremental_fold_equals_batch_fold_on_the_gpu(n_frames):
    """configs[1]'s scene (device-rendered, 640x480, 32 masks per frame): 120 frames, and the full 1000 -- the batch fold,
    the incremental fold from the first step, a switch after the clouds have grown, and the incremental fold with the
    anchor shortcut disabled: same number of instances, same points in the same order (SHA-1 of the coordinates)."""
    import torch
    import bench
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec
    L = HmsgLib()
    spec = SceneSpec(seed=1234, n_frames=n_frames, feat_dim=64, n_masks=32)
    inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
    ref = _merge_device_scene(L, spec, inp, "batch")
    assert ref[0] > 50 and ref[1] > 100 * n_frames
    modes = ["incremental", 150000] + (["incremental-noanchor"] if n_frames <= 200 else [])
    for mode in modes:
        got = _merge_device_scene(L, spec, inp, mode)
        assert got == ref, (mode, ref, got)
