"""numpy stand-ins for the third-party calls (Open3D 0.18, faiss 1.7.2) the reference makes on the
hot path, used ONLY by `gen_golden.py` in the build container to let the reference's own Python run
end to end.  They delegate to the `o3d_*` / `faiss_*` restatements in `oracle/hmsg_oracle.py`, so the
fixtures pin the reference's glue (order of operations, thresholds, torch / sklearn / scipy
semantics) but NOT the third-party internals ("parity unpinned" there, see the oracle header).
"""
from __future__ import annotations

import types

import numpy as np

from oracle import hmsg_oracle as O


class _AABB:
    def __init__(self, lo=None, hi=None, min_bound=None, max_bound=None):
        self._lo = np.asarray(lo if lo is not None else min_bound, dtype=np.float64)
        self._hi = np.asarray(hi if hi is not None else max_bound, dtype=np.float64)

    def get_box_points(self):
        # Open3D AxisAlignedBoundingBox::GetBoxPoints order
        mn, ex = self._lo, self._hi - self._lo
        return np.array([mn, mn + [ex[0], 0, 0], mn + [0, ex[1], 0], mn + [0, 0, ex[2]], self._hi,
                         mn + [0, ex[1], ex[2]], mn + [ex[0], 0, ex[2]], mn + [ex[0], ex[1], 0]])

    def get_min_bound(self):
        return self._lo

    def get_max_bound(self):
        return self._hi


class PointCloud:
    def __init__(self):
        self.points = np.zeros((0, 3))
        self.colors = np.zeros((0, 3))

    def __iadd__(self, other):
        self.points = np.concatenate([np.asarray(self.points).reshape(-1, 3), np.asarray(other.points).reshape(-1, 3)])
        # Open3D keeps colours only when both sides have them
        if len(self.colors) == len(self.points) - len(other.points) and len(other.colors) == len(other.points):
            self.colors = np.concatenate([np.asarray(self.colors).reshape(-1, 3),
                                          np.asarray(other.colors).reshape(-1, 3)])
        else:
            self.colors = np.zeros((0, 3))
        return self

    def __add__(self, other):
        out = PointCloud()
        out.points = np.asarray(self.points).copy()
        out.colors = np.asarray(self.colors).copy()
        out += other
        return out

    def transform(self, T):
        p = np.asarray(self.points, dtype=np.float64).reshape(-1, 3)
        self.points = O.o3d_transform(p, T)
        return self

    def voxel_down_sample(self, voxel_size):
        out = PointCloud()
        cols = np.asarray(self.colors) if len(self.colors) == len(self.points) else None
        p, c, _, _ = O.o3d_voxel_down_sample(np.asarray(self.points, dtype=np.float64).reshape(-1, 3), cols,
                                             voxel_size)
        out.points = p
        out.colors = c if c is not None else np.zeros((0, 3))
        return out

    def cluster_dbscan(self, eps, min_points, print_progress=False):
        return O.o3d_cluster_dbscan(np.asarray(self.points, dtype=np.float64).reshape(-1, 3), eps, min_points).tolist()

    def remove_radius_outlier(self, nb_points, radius):
        ind = O.o3d_remove_radius_outlier(np.asarray(self.points, dtype=np.float64).reshape(-1, 3), nb_points, radius)
        return self.select_by_index(ind), ind.tolist()

    def select_by_index(self, ind):
        """Open3D PointCloud::SelectByIndex: a boolean mask over the points -- the selected points come back ONCE
        each, in their original order, however often or in whatever order the index list names them."""
        out = PointCloud()
        mask = np.zeros(len(self.points), bool)
        mask[np.asarray(ind, dtype=np.int64)] = True
        out.points = np.asarray(self.points)[mask]
        if len(self.colors) == len(self.points):
            out.colors = np.asarray(self.colors)[mask]
        return out

    def get_axis_aligned_bounding_box(self):
        p = np.asarray(self.points).reshape(-1, 3)
        if p.shape[0] == 0:
            return _AABB(np.zeros(3), np.zeros(3))
        return _AABB(p.min(axis=0), p.max(axis=0))

    def crop(self, box):
        # GetPointIndicesWithinBoundingBox: min <= p <= max on every axis
        p = np.asarray(self.points).reshape(-1, 3)
        keep = np.all((p >= box.get_min_bound()) & (p <= box.get_max_bound()), axis=1)
        return self.select_by_index(np.nonzero(keep)[0])

    def is_empty(self):
        return len(self.points) == 0

    def get_center(self):
        return np.asarray(self.points).mean(axis=0)

    def has_points(self):
        return len(self.points) > 0


def _write_ply(path, pcd, *a, **k):
    """o3d.io.write_point_cloud: binary little-endian PLY with double x / y / z (colours omitted: not on the path)."""
    pts = np.asarray(pcd.points, dtype=np.float64).reshape(-1, 3)
    with open(path, "wb") as f:
        f.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty double x\nproperty double y\n"
                 "property double z\nend_header\n" % len(pts)).encode())
        f.write(pts.astype("<f8").tobytes())
    return True


def _read_ply(path, *a, **k):
    out = PointCloud()
    with open(path, "rb") as f:
        n = 0
        while True:
            line = f.readline().decode().strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line == "end_header":
                break
        out.points = np.frombuffer(f.read(n * 24), dtype="<f8").reshape(n, 3).copy()
    return out


def make_open3d():
    o3d = types.ModuleType("open3d")
    o3d.geometry = types.SimpleNamespace(PointCloud=PointCloud, AxisAlignedBoundingBox=_AABB)
    o3d.utility = types.SimpleNamespace(
        Vector3dVector=lambda a: np.array(a, dtype=np.float64).reshape(-1, 3))
    o3d.io = types.SimpleNamespace(write_point_cloud=_write_ply, read_point_cloud=_read_ply)
    o3d.visualization = types.SimpleNamespace()
    return o3d


class IndexFlatL2:
    def __init__(self, d):
        self.d = d
        self.base = np.zeros((0, d), np.float32)

    def add(self, x):
        self.base = np.concatenate([self.base, np.asarray(x, np.float32)])

    def search(self, q, k=1):
        assert k == 1
        d = O.faiss_flat_l2_nn_sqdist(np.asarray(q, np.float32), self.base)
        return d.reshape(-1, 1), np.zeros((len(q), 1), np.int64)


def make_faiss():
    f = types.ModuleType("faiss")
    f.IndexFlatL2 = IndexFlatL2
    return f


# ---- OpenCV stand-in for nav_agent/.../lio_mapping_loc/scripts/generate_depth.py (row N3) ---------------------------
def make_cv2(captured):
    """The calls generate_depth.py:125-205, 399-474 makes.  dilate / filterSpeckles are the restatements of
    oracle/lidar_depth_oracle.py (OpenCV is absent here: that part stays unpinned); imwrite stores the array in
    `captured[path]`; the overlay drawing calls are no-ops."""
    import types

    import numpy as np

    from oracle import lidar_depth_oracle as LO
    m = types.ModuleType("cv2")
    m.MORPH_RECT = 0
    m.COLORMAP_JET = 2
    m.getStructuringElement = lambda shape, ksize: np.ones((ksize[1], ksize[0]), np.uint8)

    def dilate(src, element, iterations=1):
        assert element.shape[0] == element.shape[1] and element.all()
        return LO.dilate_rect(src, element.shape[0], iterations)

    def filterSpeckles(img, newVal, maxSpeckleSize, maxDiff):
        LO.filter_speckles(img, newVal, maxSpeckleSize, maxDiff)
        return img, None

    def imwrite(path, arr):
        captured[path] = np.array(arr)
        return True

    m.dilate = dilate
    m.filterSpeckles = filterSpeckles
    m.imwrite = imwrite
    m.circle = lambda *a, **k: None
    m.applyColorMap = lambda gray, cmap: np.zeros(gray.shape + (3,), np.uint8)
    m.imread = lambda *a, **k: None
    return m
