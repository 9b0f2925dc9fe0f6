"""hmsg_config.overlap_distance_form (include/hmsg.h: HMSG_OVERLAP_DIRECT | HMSG_OVERLAP_FAISS_BLAS) -- which of faiss's two
evaluations of the squared distance find_overlapping_ratio_faiss (utils/graph_utils.py:645-662) stands for.  faiss is absent from
this image, so the switch is pinned the way the direct form is: the library (kernel simulator here, MI355X with -m gpu) against
oracle/hmsg_oracle.py with the SAME switch and the same stated order of operations, instances bit for bit -- on a scene placed tens
of metres from the origin, where the BLAS form's rounding (~|x|^2 * 2^-23) reaches the radius' scale and the two forms do disagree
on individual points (the test shows that they do, so that the switch is seen to do something)."""
import os

import numpy as np
import pytest

from oracle import hmsg_oracle as O
from tests import parity_common as PC


def _far_scene(shift=(31.0, -6.0, 27.0)):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=77, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=96, height=72,
                     n_frames=8, n_masks=8, feat_dim=16, yaw_step_deg=20.0)
    scn = SynthScene(spec)
    frames = []
    for i in range(spec.n_frames):
        fr = scn.frame(i)
        fr["pose"] = np.array(fr["pose"], np.float64)
        fr["pose"][:3, 3] += np.asarray(shift)
        frames.append(fr)
    return frames


def check_switch(L):
    frames = _far_scene()
    S = PC.stack_frames(frames)
    cfg = dict(voxel_size=0.05, init_overlap_thresh=0.75, overlap_thresh_factor=0.025, iou_thresh=0.05, merge_type="sequential", feat_dim=16)
    inst, stats = {}, {}
    for form in (0, 1):
        sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=30, outlier_radius=0.5, overlap_distance_form=form))
        sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
        sc.finalize_map()
        sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"], S["n_masks"])
        sc.fuse_frames()
        frames_pcd = [[(p, np.zeros_like(p)) for p in sc.frame_masks3d(i)] for i in range(len(frames))]
        O.OVERLAP_FORM = "faiss_blas" if form else "direct"
        try:
            ref = O.seq_merge(frames_pcd, cfg["init_overlap_thresh"], cfg["voxel_size"], cfg["iou_thresh"])
        finally:
            O.OVERLAP_FORM = "direct"
        ref = [m[0] for m in ref if m[0].shape[0] >= 10]
        sc.merge_instances()
        got = sc.instances()
        assert len(got) == len(ref) >= 3, (form, len(got), len(ref))
        for k, (g, r) in enumerate(zip(got, ref)):
            assert g.shape == r.shape and np.array_equal(g, r), (form, k)
        inst[form] = got
        if form == 0:                                    # how different the two forms are on this scene's own mask pairs
            flips = pts = 0
            clouds = [p for fr in frames_pcd for p, _ in fr if len(p) >= 20]
            r2 = np.float32((1.5 * cfg["voxel_size"]) ** 2)
            for a in clouds[:12]:
                for b in clouds[:12]:
                    if a is b:
                        continue
                    d0, d1 = O.faiss_flat_l2_nn_sqdist(a, b) < r2, O.faiss_blas_nn_sqdist(a, b) < r2
                    flips += int((d0 != d1).sum())
                    pts += len(a)
            stats = dict(points=pts, flips=flips)
        sc.close()
    print("point decisions that differ between the two forms on this scene: %d of %d" % (stats["flips"], stats["points"]))
    assert stats["flips"] > 0, "the scene does not separate the two forms"
    return inst


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_faiss_blas_form_equals_the_oracle_with_the_same_switch_on_the_simulator():
    from holoagent_amd._lib import HmsgLib
    check_switch(HmsgLib(PC.EMU_PATH))


@pytest.mark.gpu
def test_faiss_blas_form_equals_the_oracle_with_the_same_switch_gpu():
    from holoagent_amd._lib import HmsgLib
    check_switch(HmsgLib())
