"""Golden vectors of row N3 (LiDAR -> depth image): runs the REFERENCE's own generate_depth.py functions (imported from
/root/reference, build container only) on a seeded synthetic cloud and stores inputs + outputs in
tests/golden/lidar_depth.npz.  Only data travels.

    python -m oracle.refdrive.gen_golden_depth        # from the repo root

cv2 is absent here: `fake_backends.make_cv2` supplies dilate / filterSpeckles (restated, unpinned) and captures what
imwrite would have written; open3d is not touched by the functions driven here.
"""
from __future__ import annotations

import importlib.util
import os
import sys
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
SCRIPT = "/root/reference/nav_agent/humble_localization_nav2/lio_mapping_loc/scripts/generate_depth.py"


def import_reference(captured):
    from oracle.refdrive import fake_backends as FB
    sys.modules["cv2"] = FB.make_cv2(captured)
    sys.modules["open3d"] = MagicMock()
    spec = importlib.util.spec_from_file_location("ref_generate_depth", SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synth_cloud(seed=5, n=26000):
    """A 6 x 3 x 5 m room scanned from inside, a box in front of the far wall (so wall points behind it are occluded),
    a thin pole, stray points, points behind the camera; a few exact duplicates and points on pixel centres."""
    rng = np.random.Generator(np.random.PCG64(seed))
    parts = []

    def plane(o, u, v, m):
        a, b = rng.random(m), rng.random(m)
        return np.asarray(o)[None] + a[:, None] * np.asarray(u)[None] + b[:, None] * np.asarray(v)[None]
    parts.append(plane([-3, -1.5, 4.0], [6, 0, 0], [0, 3, 0], 9000))        # far wall
    parts.append(plane([-3, 1.5, 0.0], [6, 0, 0], [0, 0, 4], 5000))         # floor (y down)
    parts.append(plane([-3, -1.5, 0.0], [0, 3, 0], [0, 0, 4], 3000))        # left wall
    parts.append(plane([3, -1.5, 0.0], [0, 3, 0], [0, 0, 4], 3000))         # right wall
    parts.append(plane([-0.8, -0.2, 2.0], [1.2, 0, 0], [0, 1.4, 0], 3500))  # box front
    parts.append(plane([1.2, -1.5, 1.5], [0.05, 0, 0], [0, 3, 0], 600))     # pole
    parts.append(rng.uniform([-3, -1.5, 0.3], [3, 1.5, 4], size=(400, 3)))  # strays (speckles)
    parts.append(rng.uniform([-3, -1.5, -3], [3, 1.5, -0.1], size=(1000, 3)))  # behind the camera
    p = np.concatenate(parts)
    p = p[rng.permutation(len(p))][:n]
    p[100:110] = p[90:100]                                                  # duplicates
    return p


def main():
    captured = {}
    ref = import_reference(captured)
    W, H = 192, 144
    K = np.array([[140.0, 0, 95.5], [0, 140.0, 71.5], [0, 0, 1]])
    q = np.array([0.995, 0.02, -0.09, 0.03])
    q /= np.linalg.norm(q)
    R = ref.quaternion_to_rotation_matrix(*q)                               # world -> camera
    t = np.array([0.15, -0.1, 0.2])
    pts = synth_cloud()
    out = {}
    for tag, scale in (("s1", 1), ("s2", 2)):
        pi, pc = ref.project_points(pts, R, t, K, W, H)
        # put a handful of points exactly on pixel centres / edges of the rounding rule
        pi = pi.copy()
        pi[0, :6] = [10.5, 11.0, 0.0, W - 0.5, W - 0.50001, 37.49999999]
        pi[1, :6] = [20.5, 21.0, 0.0, 10.0, H - 0.5, 5.5]
        uvs = np.vstack((pi[0], pi[1], pc[2])).T
        flags = np.array(ref.whether_occluded_deoccfast(uvs, np.zeros((H, W, 3), np.uint8), scale, ""))
        captured.clear()
        ref.generate_occ_depth(pi, pc, W, H, "", "depth.png", "vis.png", depth_factor=1000,
                               img_input=np.zeros((H, W, 3), np.uint8), image_scale=scale)
        depth = captured["depth.png"]
        assert depth.dtype == np.uint16 and depth.shape == (H, W)
        out.update({f"{tag}_points_image": pi, f"{tag}_points_camera": pc, f"{tag}_flags": flags, f"{tag}_depth": depth})
        print(tag, "projected", pi.shape[1], "occluded", int(flags.sum()), "depth pixels", int((depth > 0).sum()))
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "lidar_depth.npz"), points=pts, R=R, t=t, K=K,
                        WH=np.array([W, H]), **out)


if __name__ == "__main__":
    main()
