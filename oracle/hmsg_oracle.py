"""CPU ORACLE for the HMSG build + retrieval path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy/scipy/sklearn restatement of the reference's algorithm for the hot path
(HorizonRobotics/HoloAgent `fsr_vln/`), every function citing the reference file:line it follows.
It is imported only by `tests/`, by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of
`bench.py` -- never by the product path (`holoagent_amd/`), which fails loudly when the HIP library
is missing.

Pinning status
--------------
* The reference's own Python (graph.py / graph_utils.py / generic.py / sam_clip_feats_extractor.py)
  was imported in the build container and driven end to end on synthetic scenes; its outputs are the
  committed fixtures `tests/golden/*.npz` (generator: `oracle/refdrive/gen_golden.py`).  The tests in
  `tests/test_oracle_golden.py` check this oracle against them.
* The arithmetic INSIDE Open3D 0.18 (voxel_down_sample, cluster_dbscan, remove_radius_outlier,
  transform, AABB), faiss 1.7.2 (IndexFlatL2) and OpenCV is third-party, un-vendored and absent from
  this image: for those calls the oracle restates the published algorithm (`o3d_*` / `faiss_*`
  functions below) and the fixture generator injects the very same restatement under the reference's
  call sites.  => "parity unpinned" for those third-party internals (output ORDER of
  voxel_down_sample is canonicalised: ascending (ix, iy, iz) voxel index).
"""
from __future__ import annotations

from collections import Counter

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree

# --------------------------------------------------------------------------------------------------
# Open3D 0.18 / faiss 1.7.2 restated semantics (third-party; see header)
# --------------------------------------------------------------------------------------------------


def o3d_voxel_keys(points: np.ndarray, voxel_size: float):
    """Open3D `PointCloud::VoxelDownSample`: voxel_min_bound = min_bound - voxel_size/2,
    index = floor((p - voxel_min_bound) / voxel_size) per axis (int)."""
    min_bound = points.min(axis=0)
    origin = min_bound - voxel_size * 0.5
    ref = (points - origin) / voxel_size
    return np.floor(ref).astype(np.int64), origin


def o3d_voxel_down_sample(points: np.ndarray, colors: np.ndarray | None, voxel_size: float):
    """Per-voxel mean of points (and colours), f64 sums in input order.  Output order is
    implementation-defined in Open3D (unordered_map) -> canonical: ascending (ix, iy, iz).
    Call sites: graph.py:348, generic.py:188, graph.py:456."""
    n = points.shape[0]
    if n == 0:
        return points.copy(), (None if colors is None else colors.copy()), np.zeros((0, 3), np.int64), None
    idx, origin = o3d_voxel_keys(points, voxel_size)
    dims = idx.max(axis=0) + 1
    lin = (idx[:, 0] * dims[1] + idx[:, 1]) * dims[2] + idx[:, 2]
    uniq, inv, cnt = np.unique(lin, return_inverse=True, return_counts=True)
    out = np.zeros((uniq.shape[0], 3))
    # np.add.at accumulates in input order (sequential f64), like AccumulatedPoint::AddPoint
    np.add.at(out, inv, points)
    out /= cnt[:, None].astype(np.float64)
    outc = None
    if colors is not None and colors.shape[0] == n:
        outc = np.zeros((uniq.shape[0], 3))
        np.add.at(outc, inv, colors)
        outc /= cnt[:, None].astype(np.float64)
    first = np.zeros(uniq.shape[0], dtype=np.int64)
    first[inv[::-1]] = np.arange(n)[::-1]
    return out, outc, idx[first], origin


def o3d_transform(points: np.ndarray, pose) -> np.ndarray:
    """Open3D `PointCloud::Transform`: p' = (T [p,1])[:3] / (T [p,1])[3].  Evaluated element-wise, left to
    right, without FMA (the same order the HIP kernels use) so results are bit-reproducible."""
    T = np.asarray(pose, dtype=np.float64)
    X, Y, Z = points[:, 0], points[:, 1], points[:, 2]
    rows = [((X * T[r, 0] + Y * T[r, 1]) + Z * T[r, 2]) + T[r, 3] for r in range(4)]
    return np.stack([rows[0] / rows[3], rows[1] / rows[3], rows[2] / rows[3]], axis=1)


def _radius_pairs(points: np.ndarray, radius: float):
    """All i<j with ||p_i - p_j||^2 < radius^2 (nanoflann radius search is strict)."""
    tree = cKDTree(points)
    pairs = tree.query_pairs(radius, output_type="ndarray")
    if pairs.shape[0]:
        d = points[pairs[:, 0]] - points[pairs[:, 1]]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        pairs = pairs[d2 < radius * radius]
    return pairs


def o3d_cluster_dbscan(points: np.ndarray, eps: float, min_points: int) -> np.ndarray:
    """Open3D `PointCloud::ClusterDBSCAN`: neighbours = strict radius search incl. self; core iff
    #nb >= min_points; clusters are grown from unlabelled core points in index order, so cluster ids
    follow the smallest core index of each core-connected component and a border point takes the
    first (= smallest-id) cluster that reaches it; noise = -1.  Call site graph_utils.py:839."""
    n = points.shape[0]
    labels = np.full(n, -1, dtype=np.int64)
    if n == 0:
        return labels
    pairs = _radius_pairs(points, eps)
    cnt = np.ones(n, dtype=np.int64)
    if pairs.shape[0]:
        cnt += np.bincount(pairs[:, 0], minlength=n) + np.bincount(pairs[:, 1], minlength=n)
    core = cnt >= min_points
    if not core.any():
        return labels
    cc = core[pairs[:, 0]] & core[pairs[:, 1]] if pairs.shape[0] else np.zeros(0, bool)
    g = coo_matrix((np.ones(int(cc.sum()), np.int8), (pairs[cc, 0], pairs[cc, 1])), shape=(n, n))
    _, comp = connected_components(g, directed=False)
    core_idx = np.nonzero(core)[0]
    # cluster id = rank of the component's smallest core index
    comp_min = {}
    for i in core_idx:
        c = comp[i]
        if c not in comp_min:
            comp_min[c] = len(comp_min)          # core_idx ascending -> first seen = smallest index
    labels[core_idx] = [comp_min[comp[i]] for i in core_idx]
    # border points: min cluster id over adjacent core points
    if pairs.shape[0]:
        big = np.iinfo(np.int64).max
        best = np.full(n, big, dtype=np.int64)
        a, b = pairs[:, 0], pairs[:, 1]
        m = core[a] & ~core[b]
        np.minimum.at(best, b[m], labels[a[m]])
        m = core[b] & ~core[a]
        np.minimum.at(best, a[m], labels[b[m]])
        upd = (~core) & (best != big)
        labels[upd] = best[upd]
    return labels


def o3d_remove_radius_outlier(points: np.ndarray, nb_points: int, radius: float) -> np.ndarray:
    """Open3D `RemoveRadiusOutliers`: keep i iff #{j: |p_i-p_j| < radius} (incl. self) > nb_points.
    Returns the kept indices (ascending).  Call site graph.py:355-358."""
    if points.shape[0] == 0:
        return np.zeros(0, np.int64)
    tree = cKDTree(points)
    # nanoflann's radius search (Open3D KDTreeFlann::SearchRadius) is STRICT, d^2 < r^2; scipy's ball query is
    # inclusive.  The two counts differ only for points that have a neighbour within one ulp of the sphere: those
    # are recounted by brute force.
    cnt = tree.query_ball_point(points, radius, return_length=True, workers=-1)
    inner = tree.query_ball_point(points, np.nextafter(radius, 0.0), return_length=True, workers=-1)
    for i in np.nonzero(cnt != inner)[0]:
        d = points - points[i]
        d2 = d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]
        cnt[i] = int(np.sum(d2 < radius * radius))
    return np.nonzero(cnt > nb_points)[0]


def _workers(n_queries: int) -> int:
    """cKDTree worker threads: all cores for big query sets, a few for medium ones, one for small ones (scipy starts a
    thread per worker and call -- on a 256-thread host that start-up dwarfs a query of a few thousand points; results do
    not depend on it)."""
    if n_queries > 200000:
        return -1
    return 16 if n_queries > 1500 else 1


def faiss_flat_l2_nn_sqdist(queries: np.ndarray, base: np.ndarray) -> np.ndarray:
    """faiss `IndexFlatL2.search(k=1)` distances: exact brute-force squared L2 on float32 data.
    Call site graph_utils.py:645-652."""
    q = queries.astype(np.float32)
    b = base.astype(np.float32)
    tree = cKDTree(b.astype(np.float64))
    _, nn = tree.query(q.astype(np.float64), k=1, workers=_workers(len(q)))
    d = q - b[nn]
    return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]   # float32


NN_TIE = "scipy"   # "scipy": cKDTree's own choice among bit-equal distances (traversal order), as the reference runs;
                   # "exact": the HIP path's rule -- the squared distance evaluated like cKDTree does
                   #          (((dx*dx) + dy*dy) + dz*dz in float64), strict minimum, and only BIT-EQUAL distances are
                   #          ties, which go to the lowest index.  Differs from "scipy" on bit-equal ties only (a point
                   #          that is the exact midpoint of two map voxels);
                   # "lowest": round 1's wide rule (candidates within 5e-10 relative are ties) -- kept for the record.


def nn_query(tree: cKDTree, pts: np.ndarray):
    """k=1 nearest neighbour, no distance cap (graph.py:409, generic.py:181, graph.py:458)."""
    if NN_TIE == "scipy" or pts.shape[0] == 0:
        return tree.query(pts, k=1, workers=_workers(len(pts)))
    k = min(8 if NN_TIE == "exact" else 4, tree.n)
    d, i = tree.query(pts, k=k, workers=_workers(len(pts)))
    if k == 1:
        return d, i
    if NN_TIE == "exact":
        diff = tree.data[i] - pts[:, None, :]
        d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        m = d2.min(axis=1, keepdims=True)
        cand = np.where(d2 == m, i, np.iinfo(np.int64).max)
        return np.sqrt(m[:, 0]), cand.min(axis=1)
    tied = d <= d[:, :1] * (1 + 5e-10)
    cand = np.where(tied, i, np.iinfo(np.int64).max)
    best = cand.min(axis=1)
    return d[:, 0], best


# --------------------------------------------------------------------------------------------------
# A1: back-projection   (dataloader/generic.py:74-138)
# --------------------------------------------------------------------------------------------------


def create_pcd(rgb, depth_u16, pose, K, scale=1000.0, mask_img=False, filter_distance=np.inf):
    """generic.py:101-138.  z is float32 (:111), x,y int64 and K f64 -> products in f64 (:122-124);
    valid pixels = depth>0 in row-major order (:117-120); whole-frame reject (:126-127); colours
    rgb/255 (:134-135); world = pose @ [p,1] / w (Open3D transform, :137).
    With mask_img=True `rgb` is the boolean mask and depth is multiplied by it (:115-116)."""
    rgb = np.asarray(rgb)
    depth = np.asarray(depth_u16)
    H, W = depth.shape[:2]
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    d = depth.astype(np.float32) / np.float32(scale)
    if mask_img:
        d = d * rgb
    m = d > 0
    xs = x[m]
    ys = y[m]
    z = d[m]
    if z.size == 0:
        return np.zeros((0, 3)), np.zeros((0, 3)), m
    X = (xs - K[0, 2]) * z / K[0, 0]
    Y = (ys - K[1, 2]) * z / K[1, 1]
    Z = z
    if Z.mean() > filter_distance:
        return np.zeros((0, 3)), np.zeros((0, 3)), np.zeros_like(m)
    pts = np.stack([X, Y, Z.astype(np.float64)], axis=1)
    cols = np.zeros((0, 3))
    if not mask_img:
        cols = rgb[m] / 255.0
    return o3d_transform(pts, pose), cols, m


# --------------------------------------------------------------------------------------------------
# A3: per-pixel feature fusion   (perception/models/sam_clip_feats_extractor.py:159-191)
# --------------------------------------------------------------------------------------------------


def _normalize_rows(x: np.ndarray, eps: float) -> np.ndarray:
    n = np.sqrt(np.sum(x * x, axis=-1, keepdims=True, dtype=np.float32))
    return x / np.maximum(n, np.float32(eps))


def fuse_mask_feats(f_g, f_masked, f_crop, masked_weight):
    """:159-175.  F_l = normalize(w_m F_masked + (1-w_m) F_crop); phi = cos(F_l, F_g) (eps 1e-6);
    w = softmax_M(phi); F_p = normalize(w F_g + (1-w) F_l).  All float32."""
    f_g = np.asarray(f_g, np.float32).reshape(1, -1)
    if np.asarray(f_masked).shape[0] == 0:       # the reference raises here (:163-164 returns a 3-tuple); no F_p rows
        return np.zeros((0, f_g.shape[1]), np.float32)
    # reference: numpy f32 arrays * python float -> stays float32 (numpy weak scalars)
    fused = (np.float32(masked_weight) * f_masked.astype(np.float32)
             + np.float32(1 - masked_weight) * f_crop.astype(np.float32)).astype(np.float32)
    f_l = _normalize_rows(fused, 1e-12)
    nl = np.maximum(np.sqrt(np.sum(f_l * f_l, -1, dtype=np.float32)), np.float32(1e-6))
    ng = np.maximum(np.sqrt(np.sum(f_g * f_g, -1, dtype=np.float32)), np.float32(1e-6))
    phi = np.sum((f_l / nl[:, None]) * (f_g / ng[:, None]), axis=-1, dtype=np.float32)
    e = np.exp(phi - phi.max())
    w = (e / e.sum(dtype=np.float32)).astype(np.float32).reshape(-1, 1)
    f_p = w * f_g + (np.float32(1) - w) * f_l
    return _normalize_rows(f_p.astype(np.float32), 1e-12)


def per_pixel_feats(masks: np.ndarray, f_p: np.ndarray) -> np.ndarray:
    """:178-190.  out[p] = sum_{i: p in mask_i} F_p[i] added in mask order, L2-normalised (zero rows
    stay zero, eps 1e-12), cast to fp16.  Returns fp16 [H*W, D]."""
    M = masks.shape[0]
    flat = masks.reshape(M, -1)
    out = np.zeros((flat.shape[1], f_p.shape[1]), dtype=np.float32)
    for i in range(M):
        out[flat[i]] += f_p[i]
    out = _normalize_rows(out, 1e-12)
    return out.astype(np.float16)


# --------------------------------------------------------------------------------------------------
# A2 + A4 + A5: global cloud, 3-D masks, per-point feature fusion   (graph.py:339-415)
# --------------------------------------------------------------------------------------------------


def pcd_denoise_dbscan(points, colors, eps, min_points):
    """graph_utils.py:827-880: keep the largest DBSCAN cluster (Counter.most_common -> first-seen
    label on ties) unless none exists or it has < 5 points (then the input is returned)."""
    labels = o3d_cluster_dbscan(points, eps, min_points)
    counter = Counter(labels.tolist())
    counter.pop(-1, None)
    if counter:
        lab, _ = counter.most_common(1)[0]
        keep = labels == lab
        if int(keep.sum()) < 5:
            return points, colors
        return points[keep], (colors[keep] if colors is not None and len(colors) == len(points) else colors)
    return points, colors


def build_global_cloud(frames, voxel_size, outlier_nb=1000, outlier_radius=1.0):
    """graph.py:339-358: concatenate all frames' clouds, voxel_down_sample, DBSCAN(0.01,100) keep
    largest (no-op at the shipped voxel sizes but executed), radius-outlier removal."""
    pts, cols = [], []
    for fr in frames:
        p, c, _ = create_pcd(fr["rgb"], fr["depth"], fr["pose"], fr["K"])
        pts.append(p)
        cols.append(c)
    pts = np.concatenate(pts)
    cols = np.concatenate(cols)
    p, c, keys, origin = o3d_voxel_down_sample(pts, cols, voxel_size)
    p, c = pcd_denoise_dbscan(p, c, 0.01, 100)
    keep = o3d_remove_radius_outlier(p, outlier_nb, outlier_radius)
    return p[keep], c[keep], dict(n_voxels=int(keys.shape[0]), origin=origin)


def create_3d_masks(masks, depth_u16, cloud_pts, cloud_cols, tree, pose, K, voxel_size, filter_distance,
                    nn_image=None):
    """generic.py:140-190: per mask, back-project the masked depth, snap every point to its nearest
    map point (k=1, no distance cap, duplicates kept), voxel_down_sample(voxel_size)."""
    out = []
    for i in range(masks.shape[0]):
        p, _, _ = create_pcd(masks[i], depth_u16, pose, K, mask_img=True, filter_distance=filter_distance)
        if p.shape[0] == 0:
            out.append((np.zeros((0, 3)), np.zeros((0, 3))))
            continue
        if nn_image is not None:      # test hook: reuse a given per-pixel NN map (tie-break independent)
            idx = nn_image[(np.asarray(depth_u16) > 0) & masks[i]]
        else:
            _, idx = nn_query(tree, p)
        sp = cloud_pts[idx]
        sc = cloud_cols[idx]
        dp, dc, _, _ = o3d_voxel_down_sample(sp, sc, voxel_size)
        out.append((dp, dc))
    return out


def fuse_frame_into_map(sum_feats, counter, f2d_fp16, depth_u16, idx):
    """graph.py:403-411.  torch advanced-index `sum[idx] += F` does not accumulate duplicates: each
    touched row receives old + F[j*] for ONE j*; single-threaded torch -> j* = last pixel in row-major
    order with that index (SURVEY hazard 7); `counter[idx] += 1` adds 1 once per touched row."""
    valid = (np.asarray(depth_u16) > 0).reshape(-1)
    F = f2d_fp16[valid]
    n = idx.shape[0]
    uniq, last_rev = np.unique(idx[::-1], return_index=True)
    last = n - 1 - last_rev
    sum_feats[uniq] += F[last].astype(np.float32)
    counter[uniq] += np.float32(1)


# --------------------------------------------------------------------------------------------------
# A6: mask merging   (utils/graph_utils.py:620-679, 883-1038)
# --------------------------------------------------------------------------------------------------


def compute_3d_bbox_iou(amin, amax, bmin, bmax):
    """graph_utils.py:883-915 (f64 AABBs)."""
    omin = np.maximum(amin, bmin)
    omax = np.minimum(amax, bmax)
    osz = np.maximum(omax - omin, 0.0)
    ov = np.prod(osz)
    va = np.prod(amax - amin)
    vb = np.prod(bmax - bmin)
    with np.errstate(divide="ignore", invalid="ignore"):
        return ov / (va + vb - ov)


OVERLAP_FORM = "direct"   # "direct": (dx*dx + dy*dy) + dz*dz per pair for every cloud (what rounds 1-4 pin; faiss below 20 queries);
                          # "faiss_blas": faiss's own switch -- from 20 queries on |x|^2 + |y|^2 - 2 x.y, clamped at 0
                          # (include/hmsg.h: HMSG_OVERLAP_FAISS_BLAS states the order of the operations)


def _fma32(a, b, c):
    """float32 fma(a, b, c) by way of float64: the product of two float32 is exact in float64, the sum is rounded to 53 bits and
    then to 24 -- equal to the single rounding of a hardware fma except on a float32 half-way case of the 53-bit sum (~2^-29)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def faiss_blas_nn_sqdist(queries: np.ndarray, base: np.ndarray) -> np.ndarray:
    """IndexFlatL2.search(k=1) distances the way faiss computes them for >= 20 queries (exhaustive_L2sqr_blas): the smallest of
    max(0, (|x|^2 + |y|^2) - 2 * x.y) over the base, float32, in the order include/hmsg.h states."""
    q = np.ascontiguousarray(queries, np.float32)
    b = np.ascontiguousarray(base, np.float32)
    n2 = lambda p: (p[:, 0] * p[:, 0] + p[:, 1] * p[:, 1]) + p[:, 2] * p[:, 2]
    qn, bn = n2(q), n2(b)
    out = np.full(len(q), np.inf, np.float32)
    two = np.float32(2.0)
    for s0 in range(0, len(q), 1024):
        qs = q[s0:s0 + 1024]
        for t0 in range(0, len(b), 8192):
            bs = b[t0:t0 + 8192]
            ip = _fma32(qs[:, None, 2], bs[None, :, 2], _fma32(qs[:, None, 1], bs[None, :, 1], qs[:, None, 0] * bs[None, :, 0]))
            d = (qn[s0:s0 + 1024, None] + bn[None, t0:t0 + 8192]) - two * ip
            np.maximum(d, np.float32(0.0), out=d)
            out[s0:s0 + 1024] = np.minimum(out[s0:s0 + 1024], d.min(axis=1))
    return out


def _nn_sqdist(queries, base):
    if OVERLAP_FORM == "faiss_blas" and queries.shape[0] >= 20:
        return faiss_blas_nn_sqdist(queries, base)
    return faiss_flat_l2_nn_sqdist(queries, base)


def find_overlapping_ratio(p1, p2, radius):
    """graph_utils.py:620-664: max over both directions of the fraction of points whose exact f32
    nearest neighbour in the other cloud is closer than radius (D < radius**2, float32 compare)."""
    if p1.shape[0] == 0 or p2.shape[0] == 0:
        return 0
    d1 = _nn_sqdist(p1, p2)
    d2 = _nn_sqdist(p2, p1)
    r2 = np.float32(radius ** 2)
    n1 = np.sum(d1 < r2)
    n2 = np.sum(d2 < r2)
    return np.max([n1 / p1.shape[0], n2 / p2.shape[0]])


def merge_point_clouds_list(clouds):
    """graph_utils.py:667-679: concatenate in list order, then keep-largest DBSCAN(eps 0.1, min 10)
    (applies to singletons too; the voxel_size argument is unused)."""
    pts = np.concatenate([c[0] for c in clouds])
    cols = np.concatenate([c[1] for c in clouds])
    return pcd_denoise_dbscan(pts, cols, 0.1, 10)


def merge_3d_masks(mask_list, overlap_threshold, radius, iou_thresh, stats=None):
    """graph_utils.py:918-956.  Empty clouds: Open3D AABB of an empty cloud is (0,0,0)-(0,0,0) ->
    volume 0 -> IoU = 0/0 = nan -> `nan > thresh` False."""
    n = len(mask_list)
    mins = [c[0].min(axis=0) if c[0].shape[0] else np.zeros(3) for c in mask_list]
    maxs = [c[0].max(axis=0) if c[0].shape[0] else np.zeros(3) for c in mask_list]
    overlap = np.zeros((n, n))
    for i in range(n):
        for j in range(i + 1, n):
            if compute_3d_bbox_iou(mins[i], maxs[i], mins[j], maxs[j]) > iou_thresh:
                overlap[i, j] = find_overlapping_ratio(mask_list[i][0], mask_list[j][0], 1.5 * radius)
                if stats is not None:
                    stats["pairs"] = stats.get("pairs", 0) + 1
    if overlap.size == 0:
        return mask_list
    graph = overlap > overlap_threshold
    ncomp, labels = connected_components(graph)
    merged = []
    for c in range(ncomp):
        members = np.where(labels == c)[0]
        merged.append(merge_point_clouds_list([mask_list[i] for i in members]))
    return merged


def seq_merge(frames_pcd, th, down_size, proxy_th, stats=None):
    """graph_utils.py:1015-1038."""
    global_masks = list(frames_pcd[0])
    for i in range(1, len(frames_pcd)):
        global_masks = merge_3d_masks(global_masks + list(frames_pcd[i]), th, down_size, proxy_th, stats)
    return merge_3d_masks(global_masks, th, down_size, proxy_th, stats)


def hierarchical_merge(frames_pcd, th, th_factor, down_size, proxy_th):
    """graph_utils.py:959-1012."""
    frames_pcd = [list(f) for f in frames_pcd]
    while len(frames_pcd) > 1:
        nxt = []
        for i in range(0, len(frames_pcd), 2):
            if i == len(frames_pcd) - 1:
                nxt.append(frames_pcd[i])
                break
            nxt.append(merge_3d_masks(frames_pcd[i] + frames_pcd[i + 1], th, down_size, proxy_th))
        frames_pcd = nxt
        if len(frames_pcd) > 1:
            th -= th_factor * (len(frames_pcd) - 2) / max(1, len(frames_pcd) - 1)
    return merge_3d_masks(frames_pcd[0], 0.75, down_size, proxy_th)


# --------------------------------------------------------------------------------------------------
# A7: per-instance pooling   (graph.py:450-491, graph_utils.py:682-728)
# --------------------------------------------------------------------------------------------------


def cosine_dbscan_labels(feats: np.ndarray, eps: float, min_samples: int) -> np.ndarray:
    """sklearn DBSCAN(metric="cosine") restated (SURVEY hazard 18): brute-force pairwise cosine
    distance 1 - x^.y^ in float32, neighbour iff d <= eps (incl. self), core iff >= min_samples, labels
    by first core index, border -> first reaching cluster."""
    from sklearn.cluster import DBSCAN
    return DBSCAN(eps=eps, min_samples=min_samples, metric="cosine").fit(feats).labels_


def feats_denoise_dbscan(feats: np.ndarray, eps=0.01, min_points=100) -> np.ndarray:
    """graph_utils.py:682-728: largest non-noise cluster (first-seen on ties) -> mean over its rows
    (np.mean axis 0, float32); a 1-row cluster is returned as [1,D]; no cluster -> mean of all rows."""
    labels = cosine_dbscan_labels(feats, eps, min_points)
    counter = Counter(labels.tolist())
    counter.pop(-1, None)
    if counter:
        lab, _ = counter.most_common(1)[0]
        sel = feats[labels == lab]
        if len(sel) > 1:
            return np.mean(sel, axis=0)
        return sel
    return np.mean(feats, axis=0)


def pool_instances(mask_pcds, cloud_pts, tree, full_feats, voxel_size, feat_dim, max_dist=0.8):
    """graph.py:450-488."""
    out = []
    for pts, cols in mask_pcds:
        dp, _, _, _ = o3d_voxel_down_sample(pts, cols, voxel_size)
        dist, idx = nn_query(tree, dp)
        valid = dist <= max_dist
        feats = np.nan_to_num(full_feats[idx[valid]])
        if feats.shape[0] == 0:
            out.append(np.zeros((1, feat_dim), dtype=full_feats.dtype))
            continue
        out.append(feats_denoise_dbscan(feats, eps=0.01, min_points=100))
    return out


# --------------------------------------------------------------------------------------------------
# create_feature_map driver   (graph.py:262-491)
# --------------------------------------------------------------------------------------------------


def create_feature_map(frames, cfg, keep_intermediates=False, stats=None):
    """frames: list of dicts (rgb, depth, pose, K, masks[M,H,W] bool, f_g, f_masked, f_crop).
    cfg: dict with voxel_size, clip_masked_weight, max_mask_distance, init_overlap_thresh,
    overlap_thresh_factor, iou_thresh, merge_type, feat_dim (+ outlier_nb, outlier_radius)."""
    vs = cfg["voxel_size"]
    D = cfg["feat_dim"]
    cloud_pts, cloud_cols, info = build_global_cloud(frames, vs, cfg.get("outlier_nb", 1000),
                                                     cfg.get("outlier_radius", 1.0))
    res = dict(cloud_pts=cloud_pts, cloud_cols=cloud_cols, info=info)
    tree = cKDTree(cloud_pts)
    V = cloud_pts.shape[0]
    counter = np.zeros((V, 1), np.float32)
    sum_feats = np.zeros((V, D), np.float32)
    frames_pcd = []
    nn_idx = []
    for fr in frames:
        f_p = fuse_mask_feats(fr["f_g"], fr["f_masked"], fr["f_crop"], cfg["clip_masked_weight"])
        f2d = per_pixel_feats(fr["masks"], f_p)
        p, _, _ = create_pcd(fr["rgb"], fr["depth"], fr["pose"], fr["K"])
        frames_pcd.append(create_3d_masks(fr["masks"], fr["depth"], cloud_pts, cloud_cols, tree, fr["pose"],
                                          fr["K"], vs, cfg["max_mask_distance"]))
        _, idx = nn_query(tree, p)
        fuse_frame_into_map(sum_feats, counter, f2d, fr["depth"], idx)
        if keep_intermediates:
            nn_idx.append(idx)
    res["counter"] = counter.copy()
    counter[counter == 0] = 1e-5
    full_feats = sum_feats / counter
    res["full_feats"] = full_feats
    if keep_intermediates:
        res["frames_pcd"] = frames_pcd
        res["nn_idx"] = nn_idx
    if cfg.get("merge_type", "sequential") == "hierarchical":
        mask_pcds = hierarchical_merge(frames_pcd, cfg["init_overlap_thresh"], cfg["overlap_thresh_factor"],
                                       vs, cfg["iou_thresh"])
    else:
        mask_pcds = seq_merge(frames_pcd, cfg["init_overlap_thresh"], vs, cfg["iou_thresh"], stats)
    mask_pcds = [m for m in mask_pcds if m[0].shape[0] >= 10]          # graph.py:445-448
    res["mask_pcds"] = mask_pcds
    res["mask_feats"] = pool_instances(mask_pcds, cloud_pts, tree, full_feats, vs, D)
    return res


# --------------------------------------------------------------------------------------------------
# A12: retrieval   (graph.py:2216-2257, 3056-3272)
# --------------------------------------------------------------------------------------------------


def query_object(text_feats, query_id, object_embs, top_k, has_negatives=True):
    """graph.py:3112-3151 on an [N',D] embedding table in candidate order.  text_feats [C,D].
    Returns (top indices into the candidate list, scores sim[query_id, idx])."""
    sim = np.dot(text_feats, object_embs.T)
    top = np.argsort(sim[query_id])[::-1][:top_k]
    if has_negatives:
        cls = np.argmax(sim, axis=0)
        mx = np.max(sim, axis=0)
        ids = np.where(cls == query_id)[0]
        if len(ids) > 0:
            top = ids[np.argsort(-mx[ids])][:top_k]
    return top, sim[query_id][top]


def query_room_label(query_feat, room_name_feats):
    """graph.py:3204-3237: all rooms within 1e-3 of the best similarity, returned in ascending
    room-list order."""
    sim = np.dot(query_feat.reshape(1, -1), room_name_feats.T)
    order = np.argsort(sim[0])[::-1]
    keep = [order[0]] + [i for i in order[1:] if np.abs(sim[0, i] - sim[0, order[0]]) < 1e-3]
    return sorted(int(i) for i in keep)


def query_room_views(query_feat, room_view_embs, room_keys, top_n=5):
    """graph.py:3247-3272: per room max over view embeddings, stable descending sort; the result dict is
    keyed by int(room_id.split("_")[-1]) so rooms of different floors with the same per-floor index
    collapse to the first occurrence (SURVEY hazard 11); first top_n keys."""
    sims = [float(np.max(np.dot(query_feat.reshape(1, -1), e.T))) for e in room_view_embs]
    order = sorted(range(len(sims)), key=lambda i: sims[i], reverse=True)
    keys = []
    for i in order:
        if room_keys[i] not in keys:
            keys.append(room_keys[i])
    return keys[:top_n]
