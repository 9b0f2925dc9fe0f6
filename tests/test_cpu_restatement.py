"""The compiled CPU restatement of the path (oracle/hmsg_cpu.cpp -- what bench.py times as `cpu_baseline`) against the
fixtures the REFERENCE's own Python produced (tests/golden/build_seq, build_hier, build_ragged: oracle/refdrive/gen_golden.py) and
against the numpy oracle on a synthetic scene: the map cloud and the merged instances bit for bit, the voxel feature map
within the fp16 knife edge, pooled instance features within 1e-5, retrieval indices exactly.  CPU only."""
import numpy as np
import pytest

from oracle import hmsg_oracle as O
from oracle.hmsg_cpu import CpuBuild
from tests import golden_io as GI


@pytest.mark.parametrize("name", ["build_seq", "build_hier", "build_ragged"])
def test_cpp_restatement_matches_reference_run(name):
    z = GI.load(name)
    frames = GI.unpack_frames(z)
    cfg = GI.unpack_cfg(z)
    b = CpuBuild(frames, cfg)
    assert np.array_equal(b.map_points(), z["ref_cloud"])
    d = np.abs(b.full_feats() - z["ref_full_feats"])
    assert (d > 1e-6).mean() < 1e-3 and d.max() < 1e-3          # (the oracle's own bound against the reference run)
    off = z["ref_mask_off"]
    inst = b.instances()
    assert len(inst) == len(off) - 1
    for i, p in enumerate(inst):
        assert np.array_equal(p, z["ref_mask_pts"][off[i]:off[i + 1]])
    np.testing.assert_allclose(b.instance_feats(), z["ref_mask_feats"], rtol=0, atol=1e-5)
    b.close()


def test_cpp_restatement_equals_oracle_on_a_synthetic_scene():
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=3, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=128, height=96,
                     n_frames=10, n_masks=8, feat_dim=64)
    sc = SynthScene(spec)
    frames = [sc.frame(i) for i in range(spec.n_frames)]
    cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=64, outlier_nb=200,
               init_overlap_thresh=0.75, overlap_thresh_factor=0.025, iou_thresh=0.05, merge_type="sequential")
    orig = O.feats_denoise_dbscan
    O.feats_denoise_dbscan = lambda f, eps=0.01, min_points=100: orig(f, eps=0.01, min_points=20)
    try:
        ref = O.create_feature_map(frames, cfg, keep_intermediates=True)
    finally:
        O.feats_denoise_dbscan = orig
    b = CpuBuild(frames, cfg, feat_dbscan_min=20)
    assert np.array_equal(b.map_points(), ref["cloud_pts"])
    masks = b.mask_clouds()
    ref_masks = [m[0] for fr in ref["frames_pcd"] for m in fr]
    assert len(masks) == len(ref_masks) and all(np.array_equal(a, c) for a, c in zip(masks, ref_masks))
    assert np.abs(b.full_feats() - ref["full_feats"]).max() <= 2.0 ** -10
    inst, ref_inst = b.instances(), [m[0] for m in ref["mask_pcds"]]
    assert len(inst) == len(ref_inst) >= 10 and all(np.array_equal(a, c) for a, c in zip(inst, ref_inst))
    feats = np.stack([np.asarray(f, np.float32).reshape(-1) for f in ref["mask_feats"]])
    np.testing.assert_allclose(b.instance_feats(), feats, rtol=0, atol=1e-5)
    # retrieval: query_hmsg_object with one negative prompt over all instances
    text, _ = sc.text_table(8)
    idx, score = b.query(text, qid=0, k=3)
    for q in range(8):
        top, s_ref = O.query_object(text[q], 0, feats.astype(np.float64), 3)
        assert [int(v) for v in idx[q] if v >= 0] == [int(t) for t in top]
        np.testing.assert_allclose(score[q][: len(top)], s_ref, rtol=0, atol=1e-6)
    b.close()
