"""Row N3 micro-benchmark: LiDAR local maps -> depth images (hmsg_lidar_depth) on cuda:0, one JSON line.
    python scripts/bench_lidar_depth.py [--frames 8] [--points 1500000] [--width 1280] [--height 720]
The CPU leg times oracle/lidar_depth_oracle.py (the reference's per-frame algorithm restated) on ONE frame."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402,F401  (torch's HIP runtime has to be loaded first)

from holoagent_amd._lib import lidar_depth  # noqa: E402
from oracle import lidar_depth_oracle as LO  # noqa: E402
from oracle.refdrive.gen_golden_depth import synth_cloud  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=8)
    ap.add_argument("--points", type=int, default=1500000)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--cpu-points", type=int, default=150000)
    a = ap.parse_args()
    W, H = a.width, a.height
    K = np.array([[0.73 * W, 0, W / 2 - 0.5], [0, 0.73 * W, H / 2 - 0.5], [0, 0, 1]])
    base = synth_cloud(3, 26000)
    rng = np.random.Generator(np.random.PCG64(1))
    clouds, poses = [], []
    for f in range(a.frames):
        idx = rng.integers(0, len(base), a.points)
        clouds.append(base[idx] + rng.normal(0, 0.01, (a.points, 3)))       # LiDAR noise around the surfaces
        poses.append(np.concatenate([np.eye(3), np.array([[0.05 * f], [0.0], [0.1]])], axis=1))
    lidar_depth(clouds[:1], poses[:1], K, W, H)                                # warm-up (allocations, code load)
    t0 = time.perf_counter()
    depth, stats, _, ms = lidar_depth(clouds, poses, K, W, H)
    wall = time.perf_counter() - t0
    # algorithmic bytes: 24 B read per input point (down-sampling), then per surviving point 24 B read + 24 B uvz +
    # 12 B sort record written and read twice per pass; per pixel 4 (inv) + 4 (tmp) + 2 (s16) + 4 (parent) + 4 (size)
    # + 4 (last) + 2 (depth) bytes, most of them touched twice
    n_in = sum(len(c) for c in clouds)
    n_ds = int(stats[:, 0].sum())
    alg = 24 * n_in * 3 + n_ds * (24 + 24 + 5 + 12 * 2 * 2) + a.frames * W * H * 2 * 24
    c0 = clouds[0][: a.cpu_points]
    t0 = time.perf_counter()
    want, _, _ = LO.lidar_depth_frame(c0, poses[0][:, :3], poses[0][:, 3], K, W, H)
    cpu_s = time.perf_counter() - t0
    got, _, _, _ = lidar_depth([c0], poses[:1], K, W, H)
    print(json.dumps({"metric": "lidar_depth_frames_per_s", "value": a.frames / (ms / 1e3), "unit": "frames/s",
                      "frames": a.frames, "image": [W, H], "points_per_frame": a.points,
                      "points_after_downsampling": n_ds // a.frames, "device_ms_per_frame": ms / a.frames,
                      "wall_ms_per_frame_incl_pcie": wall * 1e3 / a.frames,
                      "algorithmic_GBps": alg / (ms / 1e3) / 1e9,
                      "depth_pixels_per_frame": int(stats[:, 3].mean()),
                      "cpu_oracle": {"points": len(c0), "seconds": cpu_s, "cores": 1, "equal_to_gpu": bool(np.array_equal(got[0], want))}}))


if __name__ == "__main__":
    main()
