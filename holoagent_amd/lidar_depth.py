"""Host side of row N3: the reference's LiDAR -> depth-image script with the per-frame work moved to the device.

Mirrors nav_agent/humble_localization_nav2/lio_mapping_loc/scripts/generate_depth.py (same file formats, same function
names where a caller could hold on to them): read_pcd_poses :15-36, read_camera_intrinsics :211-236,
read_image_tum_trajectories :296-313, quaternion_to_rotation_matrix :317-360, load_keyframe_clouds :520-555,
process_frame :612-659, main :685-720.  Differences, all on purpose:
  * frames are processed in BATCHES through one C-ABI call (include/hmsg.h: hmsg_lidar_depth) instead of a 16-thread
    pool of per-frame numpy / OpenCV calls;
  * the overlay picture (the RGB image with coloured dots, :449-471) and the *_dilate.png debug view are not produced:
    they are visualisations, not inputs of the HMSG build; consequently the RGB image is not read and a frame
    without one is NOT skipped;
  * key-frame clouds are read by a small PCD reader (ascii / binary, x y z fields) instead of Open3D.
There is no CPU fallback: without libhmsg.so the call raises.
"""
from __future__ import annotations

import os

import numpy as np

from . import _lib

RADIUS = 4.0            # process_frame :619
VOXEL_SIZE = 0.02       # :626
DEPTH_FACTOR = 1000     # :657


def quaternion_to_rotation_matrix(qw, qx, qy, qz):
    return np.array([[1 - 2 * qy * qy - 2 * qz * qz, 2 * qx * qy - 2 * qz * qw, 2 * qx * qz + 2 * qy * qw],
                     [2 * qx * qy + 2 * qz * qw, 1 - 2 * qx * qx - 2 * qz * qz, 2 * qy * qz - 2 * qx * qw],
                     [2 * qx * qz - 2 * qy * qw, 2 * qy * qz + 2 * qx * qw, 1 - 2 * qx * qx - 2 * qy * qy]])


def read_pcd_poses(file_path):
    """Key-frame positions: lines of `tx ty tz qw qx qy qz`; the "timestamp" of a key frame is its line index."""
    poses, timestamps = [], []
    with open(file_path) as f:
        for idx, line in enumerate(f):
            parts = line.strip().split()
            if len(parts) == 7:
                poses.append([float(v) for v in parts[0:3]])
                timestamps.append(idx)
    return np.array(poses), timestamps


def read_camera_intrinsics(camera_file):
    """COLMAP cameras.txt: first data line `id model width height fx fy cx cy ...`."""
    with open(camera_file) as f:
        for line in f:
            if line.startswith("#") or line.strip() == "":
                continue
            parts = line.strip().split()
            if len(parts) < 8:
                continue
            width, height = int(parts[2]), int(parts[3])
            fx, fy, cx, cy = map(float, parts[4:8])
            return np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]]), width, height
    raise ValueError(f"Invalid camera file format in file: {camera_file}")


def read_image_tum_trajectories(images_file):
    """TUM lines `t tx ty tz qx qy qz qw` -> {f"{t:.4f}": 4x4 world -> camera}."""
    poses = {}
    with open(images_file) as f:
        for line in f:
            values = [float(v) for v in line.strip().split()]
            if len(values) != 8:
                continue
            tx, ty, tz, qx, qy, qz, qw = values[1:8]
            pose = np.eye(4)
            pose[:3, :3] = quaternion_to_rotation_matrix(qw, qx, qy, qz)
            pose[:3, 3] = [tx, ty, tz]
            poses[f"{float(values[0]):.4f}"] = pose
    return poses


def read_pcd(path):
    """x, y, z of a .pcd file (DATA ascii or binary; any extra fields are skipped)."""
    with open(path, "rb") as f:
        fields, sizes, types, counts, npts, data = [], [], [], [], 0, None
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: no DATA line")
            tok = line.decode("ascii", "replace").strip().split()
            if not tok or tok[0].startswith("#"):
                continue
            key = tok[0].upper()
            if key == "FIELDS":
                fields = tok[1:]
            elif key == "SIZE":
                sizes = [int(v) for v in tok[1:]]
            elif key == "TYPE":
                types = tok[1:]
            elif key == "COUNT":
                counts = [int(v) for v in tok[1:]]
            elif key == "POINTS":
                npts = int(tok[1])
            elif key == "DATA":
                data = tok[1].lower()
                break
        counts = counts or [1] * len(fields)
        if data == "ascii":
            arr = np.loadtxt(f, ndmin=2) if npts else np.zeros((0, sum(counts)))
            col = np.cumsum([0] + counts)
            return np.stack([arr[:, col[fields.index(a)]] for a in "xyz"], axis=1).astype(np.float64)
        if data != "binary":
            raise ValueError(f"{path}: DATA {data} is not supported")
        kinds = {("F", 4): "<f4", ("F", 8): "<f8", ("U", 1): "u1", ("U", 2): "<u2", ("U", 4): "<u4", ("I", 1): "i1",
                 ("I", 2): "<i2", ("I", 4): "<i4"}
        dt = np.dtype([(n_, kinds[(t_.upper(), s_)], (c_,)) for n_, t_, s_, c_ in zip(fields, types, sizes, counts)])
        rec = np.frombuffer(f.read(npts * dt.itemsize), dtype=dt, count=npts)
        return np.stack([rec[a][:, 0] for a in "xyz"], axis=1).astype(np.float64)


def load_keyframe_clouds(keyframe_dir, nearest_timestamps, kf_cloud_poses=None):
    """Concatenation of the key-frame clouds `<timestamp>.pcd` (they are stored in world coordinates)."""
    parts = []
    for ts in nearest_timestamps:
        path = os.path.join(keyframe_dir, f"{ts}.pcd")
        if os.path.exists(path):
            parts.append(read_pcd(path))
    return np.vstack(parts) if parts else np.empty((0, 3))


def write_depth_png(path, depth):
    """uint16 PNG, like cv2.imwrite(path, depth_map.astype(np.uint16)) (:474)."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(depth, np.uint16)).save(path)


def generate_depth_images(images_file, camera_file, keyframe_dir, traj_kf_file, output_dir, batch_frames=32,
                          radius=RADIUS, voxel_size=VOXEL_SIZE, depth_factor=DEPTH_FACTOR, image_scale=1, device_id=0,
                          lib_=None):
    """main() of the reference script: one uint16 depth PNG per image pose.  Returns {timestamp: stats row}."""
    from scipy.spatial import KDTree
    os.makedirs(output_dir, exist_ok=True)
    intrinsics, width, height = read_camera_intrinsics(camera_file)
    trajectories = read_image_tum_trajectories(images_file)
    kf_xyz, timestamps = read_pcd_poses(traj_kf_file)
    kf_tree = KDTree(kf_xyz)
    cache, report = {}, {}
    items = list(trajectories.items())
    for b0 in range(0, len(items), batch_frames):
        batch = items[b0:b0 + batch_frames]
        clouds, poses = [], []
        for ts, pose in batch:
            rotation, translation = pose[:3, :3], pose[:3, 3]
            cam2world_rot = np.linalg.inv(rotation)
            centre = -np.dot(cam2world_rot, translation)
            near = [timestamps[i] for i in kf_tree.query_ball_point(centre, r=radius)]
            for k in near:
                if k not in cache:
                    cache[k] = load_keyframe_clouds(keyframe_dir, [k])
            clouds.append(np.vstack([cache[k] for k in near]) if near else np.empty((0, 3)))
            poses.append(pose[:3, :4])
        depth, stats, _, _ = _lib.lidar_depth(clouds, poses, intrinsics, width, height, voxel_size=voxel_size,
                                              depth_factor=depth_factor, image_scale=image_scale, device_id=device_id,
                                              lib_=lib_)
        for j, (ts, _) in enumerate(batch):
            if len(clouds[j]) == 0:         # process_frame :631-634: nothing to project, no file
                continue
            write_depth_png(os.path.join(output_dir, f"{ts}.png"), depth[j])
            report[ts] = stats[j].copy()
    return report
