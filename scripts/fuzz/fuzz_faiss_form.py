"""How often does faiss's BLAS distance form (|x|^2 + |y|^2 - 2 x.y in float32, used for >= 20 queries) flip a merge decision
that the direct form (the oracle's and the HIP path's) takes?  tests/test_faiss_form_fuzz.py asserts "never" on the three
reference-made fixtures; this runs the same comparison over random synthetic scenes and prints the counts.
    python scripts/fuzz/fuzz_faiss_form.py [n_scenes=50] > profiles/rNN_faiss_form_fuzz.txt"""
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from holoagent_amd.synth import SceneSpec, SynthScene            # noqa: E402
from tests.test_faiss_form_fuzz import run_both_forms            # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    tot = dict(pairs=0, points=0, point_flips=0, pair_flips=0)
    worst = np.inf
    for k in range(n):
        rng = np.random.default_rng(1000 + k)
        spec = SceneSpec(seed=5000 + k, rooms_x=int(rng.integers(1, 3)), rooms_z=1,
                         room_size=(float(rng.uniform(3.0, 5.0)), 2.6, float(rng.uniform(3.0, 4.5))), objects_per_room=int(rng.integers(3, 7)),
                         width=128, height=96, n_frames=10, n_masks=int(rng.integers(6, 14)), feat_dim=16,
                         yaw_step_deg=float(rng.uniform(8.0, 25.0)))
        scn = SynthScene(spec)
        # the scene sits 10-40 m from the origin: the BLAS form's rounding grows with |x|^2
        shift = rng.uniform(10.0, 40.0, 3) * rng.choice([-1.0, 1.0], 3)
        frames = []
        for i in range(spec.n_frames):
            fr = scn.frame(i)
            fr["pose"] = np.array(fr["pose"], np.float64)
            fr["pose"][:3, 3] += shift
            frames.append(fr)
        cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, init_overlap_thresh=0.75, overlap_thresh_factor=0.025,
                   iou_thresh=0.05, merge_type="sequential", feat_dim=16, outlier_nb=30, outlier_radius=0.5)
        st = run_both_forms(frames, cfg)
        for key in tot:
            tot[key] += st[key]
        worst = min(worst, st["min_margin"])
        print("scene %3d (offset %6.1f m): %4d pairs, %7d point decisions, %3d differ, merge decisions that differ: %d, smallest |ratio - th| %.4f"
              % (k, float(np.linalg.norm(shift)), st["pairs"], st["points"], st["point_flips"], st["pair_flips"], st["min_margin"]), flush=True)
    print("TOTAL over %d scenes: %d pairs, %d point decisions, %d differ (%.2e), merge decisions that differ: %d, smallest |ratio - th| %.4f"
          % (n, tot["pairs"], tot["points"], tot["point_flips"], tot["point_flips"] / max(1, tot["points"]), tot["pair_flips"], worst))


if __name__ == "__main__":
    main()
