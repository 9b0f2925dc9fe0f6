"""ctypes front end of oracle/hmsg_cpu.cpp (the compiled CPU restatement of the path).  TEST INFRASTRUCTURE / CPU BASELINE
ONLY: imported by tests/ and by the cpu_baseline leg of bench.py, never by holoagent_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "hmsg_cpu.cpp")
LIB = os.path.join(HERE, "libhmsg_cpu.so")


def build(force=False):
    """g++ -O2 -fopenmp -ffp-contract=off (no fused multiply-adds: the geometry is compared bit for bit)."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.run(["g++", "-O2", "-std=c++17", "-fopenmp", "-ffp-contract=off", "-mf16c", "-shared", "-fPIC", SRC, "-o", LIB],
                       check=True)
    return LIB


class _Cfg(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("D", C.c_int32), ("outlier_nb", C.c_int32), ("feat_dbscan_min", C.c_int32),
                ("merge_hierarchical", C.c_int32), ("voxel_size", C.c_double), ("masked_weight", C.c_double), ("max_mask_distance", C.c_double),
                ("init_overlap_thresh", C.c_double), ("iou_thresh", C.c_double), ("outlier_radius", C.c_double),
                ("overlap_thresh_factor", C.c_double)]


_P = C.c_void_p


def _ptr(a):
    return a.ctypes.data_as(_P)


class CpuBuild:
    """create_feature_map (graph.py:262-491) + query_hmsg_object (graph.py:3112-3151) on the host cores."""

    def __init__(self, frames, cfg, feat_dbscan_min=100):
        self.lib = C.CDLL(build())
        L = self.lib
        L.hmsg_cpu_build.restype = _P
        for n in ("hmsg_cpu_map_size", "hmsg_cpu_num_masks", "hmsg_cpu_num_instances"):
            getattr(L, n).restype = C.c_int64
            getattr(L, n).argtypes = [_P]
        F = len(frames)
        H, W = frames[0]["depth"].shape
        D = int(np.asarray(frames[0]["f_g"]).reshape(-1).shape[0])
        M = max(max(f["masks"].shape[0] for f in frames), 1)
        self.D = D

        def pad(a, shape):
            out = np.zeros((M,) + shape, a.dtype)
            out[: a.shape[0]] = a
            return out
        rgb = np.ascontiguousarray(np.stack([np.asarray(f["rgb"], np.uint8)[..., :3] for f in frames]))
        dep = np.ascontiguousarray(np.stack([np.asarray(f["depth"]).astype(np.uint16) for f in frames]))
        pose = np.ascontiguousarray(np.stack([np.asarray(f["pose"], np.float64).reshape(16) for f in frames]))
        K = np.ascontiguousarray(np.asarray(frames[0]["K"], np.float64).reshape(9))
        masks = np.ascontiguousarray(np.stack([pad(np.asarray(f["masks"]).astype(np.uint8), (H, W)) for f in frames]))
        nm = np.array([f["masks"].shape[0] for f in frames], np.int32)
        fg = np.ascontiguousarray(np.stack([np.asarray(f["f_g"], np.float32).reshape(-1) for f in frames]))
        fm = np.ascontiguousarray(np.stack([pad(np.asarray(f["f_masked"], np.float32).reshape(-1, D), (D,)) for f in frames]))
        fc = np.ascontiguousarray(np.stack([pad(np.asarray(f["f_crop"], np.float32).reshape(-1, D), (D,)) for f in frames]))
        c = _Cfg(H, W, D, int(cfg.get("outlier_nb", 1000)), int(feat_dbscan_min), int(cfg.get("merge_type", "sequential") == "hierarchical"),
                 float(cfg["voxel_size"]), float(cfg["clip_masked_weight"]), float(cfg["max_mask_distance"]),
                 float(cfg["init_overlap_thresh"]), float(cfg["iou_thresh"]), float(cfg.get("outlier_radius", 1.0)),
                 float(cfg.get("overlap_thresh_factor", 0.025)))
        self.h = _P(L.hmsg_cpu_build(C.byref(c), F, M, _ptr(rgb), _ptr(dep), _ptr(pose), _ptr(K), _ptr(masks), _ptr(nm), _ptr(fg),
                                     _ptr(fm), _ptr(fc)))

    def map_points(self):
        out = np.empty((self.lib.hmsg_cpu_map_size(self.h), 3))
        self.lib.hmsg_cpu_get_map(self.h, _ptr(out))
        return out

    def full_feats(self):
        out = np.empty((self.lib.hmsg_cpu_map_size(self.h), self.D), np.float32)
        self.lib.hmsg_cpu_get_full_feats(self.h, _ptr(out))
        return out

    def _clouds(self, n_fn, size_fn, pts_fn):
        n = n_fn(self.h)
        sizes = np.zeros(max(n, 1), np.int64)
        size_fn(self.h, _ptr(sizes))
        sizes = sizes[:n]
        pts = np.empty((int(sizes.sum()), 3))
        pts_fn(self.h, _ptr(pts))
        o = np.concatenate([[0], np.cumsum(sizes)])
        return [pts[o[i]:o[i + 1]] for i in range(n)]

    def mask_clouds(self):
        return self._clouds(self.lib.hmsg_cpu_num_masks, self.lib.hmsg_cpu_mask_sizes, self.lib.hmsg_cpu_mask_points)

    def instances(self):
        return self._clouds(self.lib.hmsg_cpu_num_instances, self.lib.hmsg_cpu_instance_sizes, self.lib.hmsg_cpu_instance_points)

    def instance_feats(self):
        out = np.empty((self.lib.hmsg_cpu_num_instances(self.h), self.D), np.float32)
        self.lib.hmsg_cpu_instance_feats(self.h, _ptr(out))
        return out

    def query(self, text, qid=0, k=3):
        """text f32 [Q, C, D] -> (idx i32 [Q, k], score f64 [Q, k])"""
        text = np.ascontiguousarray(text, np.float32)
        Q, Cn = text.shape[0], text.shape[1]
        idx = np.empty((Q, k), np.int32)
        sc = np.empty((Q, k), np.float64)
        self.lib.hmsg_cpu_query(self.h, Q, Cn, _ptr(text), int(qid), int(k), _ptr(idx), _ptr(sc))
        return idx, sc

    def close(self):
        if self.h:
            self.lib.hmsg_cpu_free(self.h)
            self.h = None


def query_table(feats, text, qid=0, k=3):
    """query_hmsg_object over a node table on the host cores (hmsg_cpu_query_table): feats f32 [N, D], text f32 [Q, C, D]
    -> (idx i32 [Q, k], score f64 [Q, k]).  The CPU side of bench.py's queries/s."""
    L = C.CDLL(build())
    feats = np.ascontiguousarray(feats, np.float32)
    text = np.ascontiguousarray(text, np.float32)
    Q, Cn, D = text.shape
    assert feats.shape[1] == D
    idx = np.empty((Q, k), np.int32)
    sc = np.empty((Q, k), np.float64)
    L.hmsg_cpu_query_table.argtypes = [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P]
    L.hmsg_cpu_query_table(_ptr(feats), feats.shape[0], D, Q, Cn, _ptr(text), int(qid), int(k), _ptr(idx), _ptr(sc))
    return idx, sc

