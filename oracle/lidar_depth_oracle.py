"""TEST INFRASTRUCTURE -- CPU restatement of the reference's LiDAR -> depth-image step.

Reference: /root/reference/nav_agent/humble_localization_nav2/lio_mapping_loc/scripts/generate_depth.py
  project_points              :366-396   world points -> camera frame -> pixel coordinates, culls
  whether_occluded_deoccfast  :125-205   z-buffer at ROUNDED pixels, inverse depth, dilate, int16, filterSpeckles,
                                         per-point occlusion flag
  generate_occ_depth          :399-474   last-writer-wins depth image at TRUNCATED pixels, * depth_factor, uint16
  process_frame               :612-659   voxel_down_sample(0.02) of the local map, then the three above

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(holoagent_amd/csrc/hmsg_splat.hip) never does.

Pinning: the pure-Python / numpy logic of the three functions (pixel rounding vs truncation, the strict z-buffer
test against a float32 buffer, the int16 cast, the 3.0 disparity window, last-writer-wins, uint16 cast) is pinned
against the reference ITSELF: oracle/refdrive/gen_golden_depth.py imports generate_depth.py in the build container and
stores its outputs in tests/golden/lidar_depth.npz (tests/test_lidar_depth.py).  OpenCV 4.8.1 (environment.yaml:31) is
absent here, so its two calls are restated from the published algorithms and are "parity unpinned":
  cv2.dilate(src, rect k x k, iterations=4): a rectangular element with iterations > 1 is ONE pass with the element
      grown to k + 3 (k - 1) and the anchor moved to 4 * (k // 2) (imgproc/morph.cpp); border pixels are ignored
      (BORDER_CONSTANT with morphologyDefaultBorderValue) -- a max filter over offsets [-4a, 4 (k - 1 - a)], a = k // 2;
  cv2.filterSpeckles(img, 0, 1000, 1): 4-connected flood fill over non-zero pixels whose neighbouring values differ by
      at most maxDiff; regions of at most maxSpeckleSize pixels are set to newVal (calib3d/stereosgbm.cpp).  The
      relation is symmetric, so the regions are the connected components of that graph whatever the scan order.
np.dot of project_points runs in BLAS (FMA or not is the library's choice): the projection itself is compared with a
2-ulp tolerance; everything downstream is bit-exact given the projected points.
"""
from __future__ import annotations

import numpy as np

FB = 20.0              # generate_depth.py:147
ZBUF_INIT = 1000.0     # :146
MAX_SPECKLE = 1000     # :172
MAX_DIFF = 1           # :172
DISP_WINDOW = 3.0      # :203


def project_points(points, rotation, translation, intrinsics, img_width, img_height):
    """generate_depth.py:366-396 with the two np.dot calls written out (left to right, no FMA)."""
    p = np.asarray(points, np.float64)
    R = np.asarray(rotation, np.float64)
    t = np.asarray(translation, np.float64)
    K = np.asarray(intrinsics, np.float64)
    cam = np.empty((3, p.shape[0]))
    for a in range(3):
        cam[a] = ((R[a, 0] * p[:, 0] + R[a, 1] * p[:, 1]) + R[a, 2] * p[:, 2]) + t[a]
    cam = cam[:, cam[2] > 0]
    n = cam / cam[2]
    img = np.empty_like(n)
    for a in range(3):
        img[a] = (K[a, 0] * n[0] + K[a, 1] * n[1]) + K[a, 2] * n[2]
    x, y = img[0], img[1]
    ok = (x >= 0) & (x < img_width) & (y >= 0) & (y < img_height)
    return img[:, ok], cam[:, ok]


def dilate_rect(src, ksize, iterations):
    """cv2.dilate with a k x k MORPH_RECT element and default anchor / border (see the module header)."""
    a = ksize // 2
    lo, hi = iterations * a, iterations * (ksize - 1 - a)
    H, W = src.shape
    pad = np.full((H + lo + hi, W + lo + hi), -np.inf, src.dtype)
    pad[lo:lo + H, lo:lo + W] = src
    out = np.full_like(src, -np.inf)
    for dy in range(lo + hi + 1):
        for dx in range(lo + hi + 1):
            np.maximum(out, pad[dy:dy + H, dx:dx + W], out=out)
    return out


def filter_speckles(img, new_val, max_size, max_diff):
    """cv2.filterSpeckles on an int16 image, in place (see the module header)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    H, W = img.shape
    v = img.astype(np.int32)
    live = v != new_val
    idx = np.arange(H * W).reshape(H, W)
    e_h = live[:, :-1] & live[:, 1:] & (np.abs(v[:, :-1] - v[:, 1:]) <= max_diff)
    e_v = live[:-1] & live[1:] & (np.abs(v[:-1] - v[1:]) <= max_diff)
    src = np.concatenate([idx[:, :-1][e_h], idx[:-1][e_v]])
    dst = np.concatenate([idx[:, 1:][e_h], idx[1:][e_v]])
    g = coo_matrix((np.ones(len(src), np.int8), (src, dst)), shape=(H * W, H * W))
    _, lab = connected_components(g, directed=False)
    size = np.bincount(lab, minlength=lab.max() + 1)
    small = (size[lab] <= max_size).reshape(H, W) & live
    img[small] = new_val
    return img


def to_int16(a):
    """np.int16(float32 array): truncation toward zero (values beyond int16 wrap like the C cast through int32)."""
    return np.trunc(a).astype(np.int64).astype(np.int16)


def occlusion_flags(uvs, img_h, img_w, image_scale=1):
    """whether_occluded_deoccfast (generate_depth.py:125-205)."""
    uvs = np.asarray(uvs, np.float64)
    n = len(uvs)
    flag = np.zeros(n, bool)
    inv_depth = np.zeros((img_h, img_w), np.float32)
    min_depth = np.full((img_h, img_w), ZBUF_INIT, np.float32)
    x, y, z = uvs[:, 0], uvs[:, 1], uvs[:, 2]
    out = (z <= 0) | (x + 0.5 < 0) | (x + 0.5 >= img_w) | (y + 0.5 < 0) | (y + 0.5 >= img_h)
    flag[out] = True
    col = np.zeros(n, np.int64)
    row = np.zeros(n, np.int64)
    col[~out] = (x[~out] + 0.5).astype(np.int64)
    row[~out] = (y[~out] + 0.5).astype(np.int64)
    for k in np.nonzero(~out)[0]:                       # input order matters: the buffer holds float32(z)
        if float(min_depth[row[k], col[k]]) > z[k]:
            min_depth[row[k], col[k]] = z[k]
            inv_depth[row[k], col[k]] = FB / z[k]
    ksize = max(1, 4 // image_scale)
    s16 = to_int16(dilate_rect(inv_depth, ksize, 4))
    filter_speckles(s16, 0, MAX_SPECKLE, MAX_DIFF)
    ins = np.nonzero(~out)[0]
    noise = s16[row[ins], col[ins]].astype(np.float64)
    occ = (noise == 0) | (np.abs(FB / z[ins] - noise) >= DISP_WINDOW)
    flag[ins] = occ
    return flag, s16


def to_uint16(a32):
    """float32 -> uint16 as numpy does on x86-64 (truncate, wrap modulo 2^16)."""
    return (np.trunc(a32).astype(np.int64) & 0xFFFF).astype(np.uint16)


def occ_depth(points_image, points_camera, img_w, img_h, depth_factor=1000, image_scale=1):
    """generate_occ_depth (generate_depth.py:399-474) up to the image it writes (the overlay drawing is skipped)."""
    uvs = np.vstack((points_image[0], points_image[1], points_camera[2])).T
    flags, _ = occlusion_flags(uvs, img_h, img_w, image_scale)
    valid = (uvs[:, 2] > 0) & ~flags
    v = uvs[valid]
    x = v[:, 0].astype(int)
    y = v[:, 1].astype(int)
    z = v[:, 2]
    inb = (0 <= x) & (x < img_w) & (0 <= y) & (y < img_h)
    x, y, z = x[inb], y[inb], z[inb]
    depth = np.zeros((img_h, img_w), np.float32)
    for i in range(len(x)):                               # fancy assignment: the last duplicate wins
        depth[y[i], x[i]] = z[i] * depth_factor
    return to_uint16(depth), flags


def voxel_down_sample(points, voxel_size):
    """Open3D VoxelDownSample (process_frame :626-629): means of float64 sums in input order; voxels are returned in
    ascending (ix, iy, iz) order (Open3D's own order is its hash map's -- unpinned; the HIP path uses this one)."""
    from oracle.hmsg_oracle import o3d_voxel_down_sample
    return o3d_voxel_down_sample(np.asarray(points, np.float64), None, voxel_size)[0]


def lidar_depth_frame(points, rotation, translation, intrinsics, img_w, img_h, voxel_size=0.02, depth_factor=1000,
                      image_scale=1):
    """process_frame (generate_depth.py:612-659) without the file I/O."""
    pts = voxel_down_sample(points, voxel_size) if voxel_size > 0 else np.asarray(points, np.float64)
    if len(pts) == 0:
        return np.zeros((img_h, img_w), np.uint16), np.zeros(0, bool), (np.zeros((3, 0)), np.zeros((3, 0)))
    pi, pc = project_points(pts, rotation, translation, intrinsics, img_w, img_h)
    depth, flags = occ_depth(pi, pc, img_w, img_h, depth_factor, image_scale)
    return depth, flags, (pi, pc)
