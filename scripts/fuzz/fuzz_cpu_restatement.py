"""Randomised sweep: the compiled CPU restatement (oracle/hmsg_cpu.cpp) against the numpy oracle on small synthetic scenes
(different seeds, room layouts, mask counts, feature sizes, both merge types).  CPU only.

    python scripts/fuzz/fuzz_cpu_restatement.py [n_cases]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from holoagent_amd.synth import SceneSpec, SynthScene   # noqa: E402
from oracle import hmsg_oracle as O                      # noqa: E402
from oracle.hmsg_cpu import CpuBuild                     # noqa: E402

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rng = np.random.default_rng(2024)
bad = 0
for case in range(n_cases):
    D = int(rng.choice([16, 48, 64]))
    spec = SceneSpec(seed=int(rng.integers(1, 10 ** 6)), rooms_x=int(rng.integers(1, 3)), rooms_z=1, room_size=(3.6, 2.5, 3.2),
                     objects_per_room=int(rng.integers(2, 6)), width=int(rng.choice([96, 128])), height=int(rng.choice([72, 96])),
                     n_frames=int(rng.integers(5, 12)), n_masks=int(rng.integers(3, 10)), feat_dim=D)
    sc = SynthScene(spec)
    frames = [sc.frame(i) for i in range(spec.n_frames)]
    merge = "hierarchical" if case % 3 == 2 else "sequential"
    cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=D, outlier_nb=int(rng.choice([100, 200])),
               init_overlap_thresh=0.75, overlap_thresh_factor=0.025, iou_thresh=0.05, merge_type=merge)
    mp = int(rng.choice([10, 20]))
    orig = O.feats_denoise_dbscan
    O.feats_denoise_dbscan = lambda f, eps=0.01, min_points=100: orig(f, eps=0.01, min_points=mp)
    try:
        ref = O.create_feature_map(frames, cfg, keep_intermediates=True)
    finally:
        O.feats_denoise_dbscan = orig
    b = CpuBuild(frames, cfg, feat_dbscan_min=mp)
    ok_map = np.array_equal(b.map_points(), ref["cloud_pts"])
    masks, ref_masks = b.mask_clouds(), [m[0] for fr in ref["frames_pcd"] for m in fr]
    ok_masks = len(masks) == len(ref_masks) and all(np.array_equal(a, c) for a, c in zip(masks, ref_masks))
    inst, ref_inst = b.instances(), [m[0] for m in ref["mask_pcds"]]
    ok_inst = len(inst) == len(ref_inst) and all(np.array_equal(a, c) for a, c in zip(inst, ref_inst))
    dff = float(np.abs(b.full_feats() - ref["full_feats"]).max()) if ok_map else float("nan")
    df = float("nan")
    if ok_inst and inst:
        feats = np.stack([np.asarray(f, np.float32).reshape(-1) for f in ref["mask_feats"]])
        df = float(np.abs(b.instance_feats() - feats).max())
    good = ok_map and ok_masks and ok_inst and dff <= 2.0 ** -10 and (not inst or df <= 1e-5)
    bad += not good
    print("case %d  %s  F=%d %dx%d M=%d D=%d  V=%d inst=%d  map %s masks %s inst %s  full_feats %.1e  pooled %.1e  %s"
          % (case, merge, spec.n_frames, spec.width, spec.height, spec.n_masks, D, len(ref["cloud_pts"]), len(ref_inst), ok_map, ok_masks,
             ok_inst, dff, df, "ok" if good else "MISMATCH"))
    b.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
