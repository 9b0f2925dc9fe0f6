"""The sequential fold running BESIDE the fusion (hmsg_merge.hip: FoldPipe -- a worker thread with its own stream
starts folding the frames' 3-D masks while hmsg_fuse_frames is still producing them) against the fold run inside
hmsg_merge_instances (HMSG_FOLD_NOPIPE=1): the same sequence of merge_3d_masks calls (graph_utils.py:1015-1038), so
the instances have to be the same clouds, point for point, in the same order; a handle can be reset or destroyed
while its worker is still folding."""
import hashlib
import os

import numpy as np
import pytest

from tests import parity_common as PC


def _digest(inst):
    h = hashlib.sha1()
    for a in inst:
        h.update(np.ascontiguousarray(a).tobytes())
    return len(inst), sum(len(a) for a in inst), h.hexdigest()


def _host_scene(L, frames, feat_dim, nopipe, split=False, env=None):
    os.environ.pop("HMSG_FOLD_NOPIPE", None)
    if nopipe:
        os.environ["HMSG_FOLD_NOPIPE"] = "1"
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        sc = PC.make_scene(L, frames, dict(feat_dim=feat_dim, outlier_nb_points=60, feat_dbscan_min=8))
        S = PC.stack_frames(frames)
        sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
        sc.finalize_map()
        n = len(frames)
        cuts = [0, n // 2, n] if split else [0, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sc.add_frame_features(a, S["masks"][a:b], S["f_g"][a:b], S["f_masked"][a:b], S["f_crop"][a:b], S["n_masks"][a:b])
        sc.fuse_frames()
        sc.merge_instances()
        sc.pool_instances()
        inst, feats = sc.instances(), sc.instance_feats()
        sc.close()
        return inst, feats
    finally:
        os.environ.pop("HMSG_FOLD_NOPIPE", None)
        for k in (env or {}):
            os.environ.pop(k, None)


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_pipelined_fold_equals_fold_in_merge_on_the_simulator():
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib(PC.EMU_PATH)
    spec = SceneSpec(seed=5, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                     n_frames=6, n_masks=5, feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    ref, ref_f = _host_scene(L, frames, 16, nopipe=True)
    assert len(ref) >= 3
    got, got_f = _host_scene(L, frames, 16, nopipe=False, split=True)
    assert len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    assert np.array_equal(got_f, ref_f)
    # a handle that is reset / destroyed while its worker still has frames to fold
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=60, feat_dbscan_min=8))
    S = PC.stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.reset()
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
    sc.fuse_frames()
    sc.close()


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_inherited_overlap_grids_on_the_simulator(capfd):
    """A merged cloud whose first member came through its DBSCAN whole keeps that member's overlap grid and indexes only what it
    gained (hmsg_merge.hip: Cloud::nb, a base grid + a delta grid): same instances as with every merged cloud indexed afresh."""
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib(PC.EMU_PATH)
    spec = SceneSpec(seed=11, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                     n_frames=8, n_masks=5, feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    ref, ref_f = _host_scene(L, frames, 16, nopipe=True, env={"HMSG_DEBUG_NO_GRID_INHERIT": "1", "HMSG_DEBUG_TIMING": "1"})
    err0 = capfd.readouterr().err
    got, got_f = _host_scene(L, frames, 16, nopipe=True, env={"HMSG_DEBUG_GRID_DELTA_ALWAYS": "1", "HMSG_DEBUG_TIMING": "1"})
    err1 = capfd.readouterr().err
    import re
    assert re.search(r"overlap grids: \d+ over whole clouds \(\d+ points\), 0 delta grids", err0), err0
    m = re.search(r"overlap grids: \d+ over whole clouds \(\d+ points\), (\d+) delta grids", err1)
    assert m and int(m.group(1)) >= 3, err1
    assert len(ref) >= 3 and len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    assert np.array_equal(got_f, ref_f)


def _device_scene(L, spec, inp, nopipe, chunks=1, env=None):
    from holoagent_amd._lib import Scene
    os.environ.pop("HMSG_FOLD_NOPIPE", None)
    if nopipe:
        os.environ["HMSG_FOLD_NOPIPE"] = "1"
    for k, v in (env or {}).items():
        os.environ[k] = v
    try:
        sc = Scene(lib_=L, height=spec.height, width=spec.width, max_frames=spec.n_frames, max_masks=spec.n_masks, feat_dim=spec.feat_dim)
        sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
        sc.finalize_map()
        F = spec.n_frames
        cuts = [round(i * F / chunks) for i in range(chunks + 1)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sc.add_frame_features(a, inp["masks"][a:b], inp["f_g"][a:b], inp["f_masked"][a:b], inp["f_crop"][a:b])
        sc.fuse_frames()
        sc.merge_instances()
        sc.pool_instances()
        d = _digest(sc.instances()), hashlib.sha1(np.ascontiguousarray(sc.instance_feats()).tobytes()).hexdigest()
        sc.close()
        return d
    finally:
        os.environ.pop("HMSG_FOLD_NOPIPE", None)
        for k in (env or {}):
            os.environ.pop(k, None)


@pytest.mark.gpu
def test_pipelined_fold_equals_fold_in_merge_on_the_gpu():
    """configs[1]'s scene (device-rendered, 640x480, 32 masks per frame), 300 frames = 5 fusion batches: the fold beside
    the fusion (features handed over in one piece, and in three), and the fold inside hmsg_merge_instances -- same instances (SHA-1
    of the coordinates), same pooled features."""
    import torch
    import bench
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec
    L = HmsgLib()
    spec = SceneSpec(seed=1234, n_frames=300, feat_dim=64, n_masks=32)
    inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
    ref = _device_scene(L, spec, inp, nopipe=True)
    assert ref[0][0] > 50
    assert _device_scene(L, spec, inp, nopipe=False) == ref
    assert _device_scene(L, spec, inp, nopipe=False, chunks=3) == ref
    assert _device_scene(L, spec, inp, nopipe=False) == ref          # (handles re-use the allocator cache of both threads)


@pytest.mark.gpu
def test_inherited_overlap_grids_on_the_gpu():
    """configs[1]'s scene, 300 frames: merged clouds that keep their first member's overlap grid and index only what they gained
    (the default), against every merged cloud indexed afresh (HMSG_DEBUG_NO_GRID_INHERIT=1) and against delta grids whatever
    the gain (HMSG_DEBUG_GRID_DELTA_ALWAYS=1): same instances, same pooled features."""
    import torch
    import bench
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec
    L = HmsgLib()
    spec = SceneSpec(seed=77, n_frames=300, feat_dim=64, n_masks=32)
    inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
    ref = _device_scene(L, spec, inp, nopipe=True, env={"HMSG_DEBUG_NO_GRID_INHERIT": "1"})
    assert ref[0][0] > 50
    assert _device_scene(L, spec, inp, nopipe=True) == ref
    assert _device_scene(L, spec, inp, nopipe=False) == ref
    assert _device_scene(L, spec, inp, nopipe=True, env={"HMSG_DEBUG_GRID_DELTA_ALWAYS": "1"}) == ref


@pytest.mark.gpu
def test_cropped_anchor_members_on_the_gpu():
    """configs[1]'s scene, 300 frames: DBSCAN batches that bin only the part of an anchor member within 2 eps of the component's
    other members (SegDesc::forced, the default) against batches that bin every anchor whole (HMSG_DEBUG_NO_CROP=1) and against
    the fold without the anchor hint at all (HMSG_DEBUG_NOANCHOR=1): same instances, same pooled features."""
    import torch
    import bench
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec
    L = HmsgLib()
    spec = SceneSpec(seed=4321, n_frames=300, feat_dim=64, n_masks=32)
    inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
    ref = _device_scene(L, spec, inp, nopipe=True, env={"HMSG_DEBUG_NO_CROP": "1"})
    assert ref[0][0] > 50
    assert _device_scene(L, spec, inp, nopipe=True) == ref
    assert _device_scene(L, spec, inp, nopipe=False) == ref
    assert _device_scene(L, spec, inp, nopipe=True, env={"HMSG_DEBUG_NOANCHOR": "1"}) == ref
    # round 5: anchor members that stay where they are in the pool (SegDesc::out_mode 2, the default) against dense copies
    assert _device_scene(L, spec, inp, nopipe=True, env={"HMSG_DEBUG_NO_INPLACE": "1"}) == ref


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_cropped_anchor_members_on_the_simulator(capfd):
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib(PC.EMU_PATH)
    spec = SceneSpec(seed=21, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                     n_frames=8, n_masks=5, feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    ref, ref_f = _host_scene(L, frames, 16, nopipe=True, env={"HMSG_DEBUG_NO_CROP": "1", "HMSG_DEBUG_TIMING": "1"})
    err0 = capfd.readouterr().err
    got, got_f = _host_scene(L, frames, 16, nopipe=True, env={"HMSG_DEBUG_TIMING": "1"})
    err1 = capfd.readouterr().err
    import re
    assert re.search(r"cropped anchor member: 0 ", err0), err0
    m = re.search(r"cropped anchor member: (\d+) ", err1)
    assert m and int(m.group(1)) >= 2, err1
    assert len(ref) >= 3 and len(got) == len(ref) and all(np.array_equal(a, b) for a, b in zip(got, ref))
    assert np.array_equal(got_f, ref_f)
    # round 5: some of those segments ran IN PLACE (the anchor member neither gathered nor copied, the kept rest appended behind
    # it); the same fold with every output a dense copy gives the same instances
    m = re.search(r"(\d+) of them in place", err1)
    assert m and int(m.group(1)) >= 1, err1
    dense, dense_f = _host_scene(L, frames, 16, nopipe=True, env={"HMSG_DEBUG_NO_INPLACE": "1", "HMSG_DEBUG_TIMING": "1"})
    err2 = capfd.readouterr().err
    assert re.search(r" 0 of them in place", err2), err2
    assert len(dense) == len(ref) and all(np.array_equal(a, b) for a, b in zip(dense, ref)) and np.array_equal(dense_f, ref_f)


_SPLIT_SNIPPET = """
import sys, json
sys.path.insert(0, {root!r})
from tests import parity_common as PC
from tests.test_fold_pipeline import _host_scene, _digest
from holoagent_amd._lib import HmsgLib
from holoagent_amd.synth import SceneSpec, SynthScene
L = HmsgLib(PC.EMU_PATH)
spec = SceneSpec(seed=21, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=64, height=48,
                 n_frames=8, n_masks=5, feat_dim=16)
scn = SynthScene(spec)
frames = [scn.frame(i) for i in range(spec.n_frames)]
inst, feats = _host_scene(L, frames, 16, nopipe=True)
import hashlib
print("DIGEST", json.dumps([list(_digest(inst)), hashlib.sha1(feats.tobytes()).hexdigest()]))
"""


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_three_launch_compaction_equals_the_fused_one_on_the_simulator():
    """HMSG_DB_COMPACT_SPLIT=1 (keep flags, scan and scatter as three launches: the form before round 4, kept for comparison runs)
    is read ONCE per process and decides both the compaction kernels and whether anchor members may be cropped / extended in place
    (the legacy kernels know neither).  Two processes, the switch on and off, same scene: same instances, same pooled features."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for split in (False, True):
        env = dict(os.environ)
        env.pop("HMSG_DB_COMPACT_SPLIT", None)
        if split:
            env["HMSG_DB_COMPACT_SPLIT"] = "1"
        r = subprocess.run([sys.executable, "-c", _SPLIT_SNIPPET.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1][7:]))
    assert out[0] == out[1] and out[0][0][0] >= 3, out


_OV_GPU_SNIPPET = """
import sys, json
sys.path.insert(0, {root!r})
import torch
import bench
from tests.test_fold_pipeline import _device_scene
from holoagent_amd._lib import HmsgLib
from holoagent_amd.synth import SceneSpec
L = HmsgLib()
spec = SceneSpec(seed=1234, n_frames=300, feat_dim=64, n_masks=32)
inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
d = _device_scene(L, spec, inp, nopipe=True)
print("DIGEST", json.dumps([list(d[0]), d[1]]))
"""


def _overlap_launch_forms(snippet):
    """Round 6's forms of the overlap test -- X's points from X's own cell-sorted copies, the second direction planned on the device
    (k_ov_query_second), the workgroup table of the first direction only in the packed upload (the default) -- against the second
    direction as a workgroup per chunk of every larger cloud (HMSG_OV_LEGACY_SECOND=1), X's points from the pool
    (HMSG_OV_POOL_ORDER=1) and both directions in one launch (HMSG_OV_ONE_LAUNCH=1).  The switches are read once per process: four
    processes, one scene, the same instances and pooled features from all."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    keys = ("HMSG_OV_LEGACY_SECOND", "HMSG_OV_POOL_ORDER", "HMSG_OV_ONE_LAUNCH")
    for over in ({}, {"HMSG_OV_LEGACY_SECOND": "1"}, {"HMSG_OV_POOL_ORDER": "1", "HMSG_OV_LEGACY_SECOND": "1"}, {"HMSG_OV_ONE_LAUNCH": "1"}):
        env = dict(os.environ)
        for k in keys:
            env.pop(k, None)
        env.update(over)
        r = subprocess.run([sys.executable, "-c", snippet.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1][7:]))
    assert out[0] == out[1] == out[2] == out[3], out
    return out[0]


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_overlap_launch_forms_on_the_simulator():
    assert _overlap_launch_forms(_SPLIT_SNIPPET)[0][0] >= 3


@pytest.mark.gpu
def test_overlap_launch_forms_on_the_gpu():
    """configs[1]'s scene, 300 frames (see _overlap_launch_forms)."""
    assert _overlap_launch_forms(_OV_GPU_SNIPPET)[0][0] > 50
