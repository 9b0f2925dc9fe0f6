"""ctypes binding of the C ABI in include/hmsg.h.

The product path loads ONLY `holoagent_amd/libhmsg.so` (hand-written HIP for gfx950, built by
`__graft_entry__.build()` / `make -C holoagent_amd/csrc`) and raises if it is missing -- there is no
CPU fallback.  (`HmsgLib(path)` with an explicit path exists so the test-suite can drive the kernel
simulator build of the same sources, tests/emu/.)
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libhmsg.so")


class HmsgConfig(C.Structure):
    _fields_ = [
        ("device_id", C.c_int32), ("feat_dim", C.c_int32), ("height", C.c_int32), ("width", C.c_int32),
        ("max_frames", C.c_int32), ("max_masks", C.c_int32),
        ("voxel_size", C.c_double), ("depth_scale", C.c_double), ("init_overlap_thresh", C.c_double),
        ("overlap_thresh_factor", C.c_double), ("iou_thresh", C.c_double), ("clip_masked_weight", C.c_double),
        ("max_mask_distance", C.c_double), ("merge_type", C.c_int32), ("outlier_nb_points", C.c_int32),
        ("outlier_radius", C.c_double), ("pool_max_dist", C.c_double), ("feat_dbscan_eps", C.c_double),
        ("feat_dbscan_min", C.c_int32), ("merge_dbscan_eps", C.c_double), ("merge_dbscan_min", C.c_int32),
        ("min_instance_points", C.c_int32), ("skip_frames", C.c_int32), ("depth_cut", C.c_double), ("grid_resolution", C.c_double),
        ("overlap_distance_form", C.c_int32),
    ]


class HmsgNode(C.Structure):          # include/hmsg.h: hmsg_node
    _fields_ = [("instance", C.c_int32), ("floor", C.c_int32), ("room", C.c_int32), ("counter", C.c_int32),
                ("label", C.c_int32), ("n_points", C.c_int64)]


class HmsgDepthParams(C.Structure):    # include/hmsg.h: hmsg_depth_params
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("image_scale", C.c_int32), ("fx", C.c_double),
                ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("voxel_size", C.c_double),
                ("depth_factor", C.c_double)]


class HmsgObjectRecord(C.Structure):   # include/hmsg.h: hmsg_object_record
    _fields_ = [("instance", C.c_int32), ("file_stem", C.c_char_p), ("object_id_json", C.c_char_p),
                ("room_id_json", C.c_char_p), ("name_json", C.c_char_p), ("view_ids_json", C.c_char_p),
                ("best_view_id_json", C.c_char_p)]


class HmsgJsonField(C.Structure):      # include/hmsg.h: hmsg_json_field
    _fields_ = [("key", C.c_char_p), ("kind", C.c_int32), ("ndim", C.c_int32), ("n0", C.c_int64), ("n1", C.c_int64), ("data", C.c_void_p)]


class HmsgGraphParams(C.Structure):    # include/hmsg.h: hmsg_graph_params
    _fields_ = [("num_views", C.c_int32), ("kmeans_n_init", C.c_int32), ("kmeans_max_iter", C.c_int32), ("kmeans_seed", C.c_uint32),
                ("skip_frames", C.c_int32), ("image_width", C.c_int32), ("image_height", C.c_int32), ("min_visible_ratio", C.c_double),
                ("max_view_depth", C.c_double), ("host_threads", C.c_int32), ("merge_objects_graph", C.c_int32), ("reserved_", C.c_int32)]


class HmsgGraphCounts(C.Structure):    # include/hmsg.h: hmsg_graph_counts
    _fields_ = [("floors", C.c_int32), ("rooms", C.c_int32), ("views", C.c_int32), ("objects", C.c_int32), ("edges", C.c_int64),
                ("view_object_links", C.c_int64), ("begin_ms", C.c_double), ("finish_ms", C.c_double), ("kmeans_wait_ms", C.c_double)]


class HmsgGraphObject(C.Structure):    # include/hmsg.h: hmsg_graph_object
    _fields_ = [("object_id", C.c_char * 48), ("name", C.c_char * 80), ("room", C.c_int32), ("instance", C.c_int32), ("label", C.c_int32),
                ("n_views", C.c_int32), ("best_view", C.c_int32)]


class HmsgGraphRoom(C.Structure):      # include/hmsg.h: hmsg_graph_room
    _fields_ = [("room_id", C.c_char * 32), ("name", C.c_char * 80), ("floor", C.c_int32), ("n_vertices", C.c_int64), ("n_points", C.c_int64),
                ("n_embeddings", C.c_int32), ("n_sample_images", C.c_int32), ("n_objects", C.c_int32), ("n_views", C.c_int32)]


class HmsgError(RuntimeError):
    pass


_P = C.c_void_p
_SIGS = {
    "hmsg_default_config": (None, [C.POINTER(HmsgConfig)]),
    "hmsg_config_size": (C.c_size_t, []),
    "hmsg_create": (C.c_int, [C.POINTER(HmsgConfig), C.POINTER(_P)]),
    "hmsg_destroy": (None, [_P]),
    "hmsg_last_error": (C.c_char_p, [_P]),
    "hmsg_version": (C.c_char_p, []),
    "hmsg_release_cached_memory": (None, []),
    "hmsg_reset": (C.c_int, [_P]),
    "hmsg_set_profiling": (C.c_int, [_P, C.c_int32]),
    "hmsg_profile_count": (C.c_int32, [_P]),
    "hmsg_profile_entry": (C.c_int, [_P, C.c_int32, C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double)]),
    "hmsg_synth_render": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, _P,
                                    C.c_int32, _P, _P, C.c_double, C.c_uint64, _P, _P, _P, _P]),
    "hmsg_add_frames": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P]),
    "hmsg_finalize_map": (C.c_int, [_P]),
    "hmsg_map_size": (C.c_int64, [_P]),
    "hmsg_num_tie_queries": (C.c_int64, [_P]),
    "hmsg_map_size_unfiltered": (C.c_int64, [_P]),
    "hmsg_get_map_points": (C.c_int, [_P, _P, _P]),
    "hmsg_add_frame_features": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "hmsg_get_frame_num_masks": (C.c_int32, [_P, C.c_int32]),
    "hmsg_fuse_frames": (C.c_int, [_P]),
    "hmsg_get_map_feats": (C.c_int, [_P, _P, _P]),
    "hmsg_get_feature_sums": (C.c_int, [_P, _P, _P]),
    "hmsg_set_feature_sums": (C.c_int, [_P, _P, _P]),
    "hmsg_get_frame_nn": (C.c_int, [_P, C.c_int32, _P]),
    "hmsg_get_frame_fp": (C.c_int, [_P, C.c_int32, _P]),
    "hmsg_get_frame_mask_sizes": (C.c_int, [_P, C.c_int32, _P]),
    "hmsg_get_frame_mask_points": (C.c_int, [_P, C.c_int32, _P]),
    "hmsg_merge_instances": (C.c_int, [_P]),
    "hmsg_set_frame_window": (C.c_int, [_P, C.c_int32]),
    "hmsg_merge_tree_local": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "hmsg_merge_tree_join": (C.c_int, [_P, C.c_int32, _P, _P, C.c_double, C.c_int32]),
    "hmsg_num_instances": (C.c_int64, [_P]),
    "hmsg_get_instance_sizes": (C.c_int, [_P, _P]),
    "hmsg_get_instance_points": (C.c_int, [_P, _P]),
    "hmsg_get_instance_boxes": (C.c_int, [_P, _P]),
    "hmsg_denoise_instances": (C.c_int, [_P, C.c_double, C.c_int32]),
    "hmsg_instance_room_share": (C.c_int, [_P, C.c_int32, _P, _P, C.c_double, _P]),
    "hmsg_voxel_down_sample": (C.c_int, [_P, _P, C.c_int64, C.c_double, _P, _P]),
    "hmsg_pool_instances": (C.c_int, [_P]),
    "hmsg_get_instance_feats": (C.c_int, [_P, _P]),
    "hmsg_build_object_nodes": (C.c_int, [_P, C.c_int32, _P, _P, C.c_int32, _P, _P, _P, C.c_int32, _P]),
    "hmsg_num_nodes": (C.c_int64, [_P]),
    "hmsg_get_nodes": (C.c_int, [_P, _P, _P]),
    "hmsg_index_from_nodes": (C.c_int, [_P, C.POINTER(_P)]),
    "hmsg_room_clouds": (C.c_int, [_P, C.c_double, C.c_double, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "hmsg_graph_default_params": (None, [C.POINTER(HmsgGraphParams)]),
    "hmsg_build_graph": (C.c_int, [_P, C.POINTER(HmsgGraphParams), C.c_int32, _P, _P, _P, _P, C.c_int32, _P, _P, C.POINTER(_P)]),
    "hmsg_graph_begin": (C.c_int, [_P, C.POINTER(HmsgGraphParams), C.c_int32, _P, _P, _P, _P, C.POINTER(_P)]),
    "hmsg_graph_finish": (C.c_int, [_P, C.c_int32, _P, _P]),
    "hmsg_graph_destroy": (None, [_P]),
    "hmsg_graph_last_error": (C.c_char_p, [_P]),
    "hmsg_graph_get_counts": (C.c_int, [_P, C.POINTER(HmsgGraphCounts)]),
    "hmsg_graph_get_edges": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "hmsg_graph_get_objects": (C.c_int, [_P, _P, C.c_int64]),
    "hmsg_graph_get_rooms": (C.c_int, [_P, _P, C.c_int64]),
    "hmsg_graph_get_room_vertices": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "hmsg_graph_get_room_embeddings": (C.c_int, [_P, C.c_int32, _P, C.c_int64]),
    "hmsg_graph_to_json": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "hmsg_save": (C.c_int, [_P, C.c_char_p]),
    "hmsg_load": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_P)]),
    "hmsg_graph_index": (C.c_int, [_P, _P, C.POINTER(_P)]),
    "hmsg_graph_query": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "hmsg_kmeans": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, _P, _P, _P, _P]),
    "hmsg_room_camera_distances": (C.c_int, [_P, C.c_int32, C.c_int64, _P, _P]),
    "hmsg_object_views": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int64, _P, _P, C.c_double, C.c_double, _P, _P]),
    "hmsg_segment_floors": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "hmsg_segment_rooms": (C.c_int, [_P, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, _P, C.c_int64, C.POINTER(C.c_int32),
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    "hmsg_points_min_dist_2d": (C.c_int, [C.c_int32, C.c_int32, _P, _P, C.c_int64, _P, _P]),
    "hmsg_lidar_depth": (C.c_int, [C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P, _P, _P]),
    "hmsg_crop_resize_batch": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_double, C.c_int32, _P, _P, _P]),
    "hmsg_save_objects": (C.c_int, [_P, C.c_char_p, C.c_int64, _P, C.c_int32]),
    "hmsg_graph_allgather_index": (C.c_int, [_P, _P, _P, C.POINTER(_P), _P, _P, _P]),
    "hmsg_comm_send": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32]),
    "hmsg_comm_recv": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32]),
    "hmsg_merge_tree_sharded": (C.c_int, [_P, _P, C.c_int32, C.POINTER(C.c_int32)]),
    "hmsg_merge_room_objects": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_double, C.c_double, C.POINTER(C.c_int32), _P, _P, C.c_int32]),
    "hmsg_write_json": (C.c_int, [C.c_char_p, C.c_int32, _P]),
    "hmsg_write_ply": (C.c_int, [C.c_char_p, _P, C.c_int64]),
    "hmsg_read_json_numbers": (C.c_int, [C.c_char_p, C.c_char_p, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "hmsg_assign_cameras_to_rooms": (C.c_int, [_P, C.c_int64, C.c_int32, _P, C.c_double, C.c_double, _P, _P, _P]),
    "hmsg_pick_representative_views": (C.c_int, [_P, C.c_int64, C.c_int32, _P, _P, C.c_int32, _P, C.POINTER(C.c_int32)]),
    "hmsg_graph_edges": (C.c_int, [C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_int64)]),
    "hmsg_test_format_doubles": (C.c_int64, [_P, C.c_int64, _P, C.c_int64]),
    "hmsg_test_allocator_carving": (C.c_int, [C.c_int32, C.c_int32]),
    "hmsg_test_dbscan": (C.c_int, [_P, C.c_int32, _P, C.c_double, C.c_int32, _P, _P, _P, _P, _P]),
    "hmsg_index_load_objects": (C.c_int, [C.c_int32, C.c_char_p, C.c_int64, _P, _P, C.c_int32, C.POINTER(_P), C.POINTER(C.c_int32)]),
    "hmsg_index_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int64, _P, C.c_int32, _P, C.POINTER(_P)]),
    "hmsg_index_destroy": (None, [_P]),
    "hmsg_index_last_error": (C.c_char_p, [_P]),
    "hmsg_index_set_profiling": (C.c_int, [_P, C.c_int32]),
    "hmsg_index_profile": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "hmsg_query_objects": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "hmsg_similarity": (C.c_int, [_P, C.c_int32, _P, _P]),
    "hmsg_index_set_hierarchy": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "hmsg_query_hier": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "hmsg_comm_unique_id": (C.c_int, [_P]),
    "hmsg_comm_create": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "hmsg_comm_destroy": (None, [_P]),
    "hmsg_comm_last_error": (C.c_char_p, [_P]),
    "hmsg_allgather_nodes": (C.c_int, [_P, _P, C.c_int32, C.POINTER(_P), _P, _P]),
    "hmsg_allreduce_feature_sums": (C.c_int, [_P, _P]),
    "hmsg_test_sort_pairs": (C.c_int, [_P, _P, C.c_int64, C.c_int32]),
    "hmsg_test_repeat_add": (C.c_int, [_P, _P, _P, _P, C.c_int64]),
    "hmsg_test_ckdtree": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)


class HmsgLib:
    def __init__(self, path: str | None = None):
        path = path or LIB_PATH
        if not os.path.exists(path):
            raise HmsgError(
                f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
                "or make -C holoagent_amd/csrc).  There is no CPU fallback.")
        self.path = path
        self.c = C.CDLL(path)
        for name, (res, args) in _SIGS.items():
            fn = getattr(self.c, name)
            fn.restype = res
            fn.argtypes = args
        if self.c.hmsg_config_size() != C.sizeof(HmsgConfig):     # (a stale struct would let hmsg_default_config write past it)
            raise HmsgError("struct hmsg_config of %s has %d bytes, the binding's HmsgConfig %d: rebuild the library or update "
                            "holoagent_amd/_lib.py" % (path, self.c.hmsg_config_size(), C.sizeof(HmsgConfig)))

    def default_config(self, **over) -> HmsgConfig:
        cfg = HmsgConfig()
        self.c.hmsg_default_config(C.byref(cfg))
        for k, v in over.items():
            if not hasattr(cfg, k):
                raise HmsgError(f"unknown config key {k}")
            setattr(cfg, k, v)
        return cfg


_default = None


def lib() -> HmsgLib:
    global _default
    if _default is None:
        _default = HmsgLib()
    return _default


def _ptr(a):
    """numpy array / torch tensor (host or device) / int -> void*"""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):
        assert a.is_contiguous()
        # A device tensor may still be in the making on torch's current stream (an all-reduce under the nccl backend
        # only makes that stream wait; a dtype conversion is a kernel).  The library works on its own non-blocking
        # stream, which has no ordering against torch's: drain torch's stream before the pointer leaves.  (In the other
        # direction every entry point that writes into a caller's buffer waits for its own stream before returning.)
        if getattr(a, "is_cuda", False):
            import torch
            torch.cuda.current_stream(a.device).synchronize()
        return C.c_void_p(a.data_ptr())
    raise TypeError(type(a))


class Comm:
    """RCCL communicator behind the C ABI (include/hmsg.h: hmsg_comm_*).  `Comm.single()` = one rank, no communicator (every
    collective is the identity); `Comm.from_torch(device_id)` = the ranks of an initialised torch.distributed job: rank 0
    makes the id, the process group carries the 128 bytes to the others, every rank joins with its own GPU."""

    def __init__(self, h, rank, world, lib_):
        self.h, self.rank, self.world, self.L = h, rank, world, lib_

    @classmethod
    def single(cls, device_id=0, lib_=None):
        L = lib_ or lib()
        h = _P()
        rc = L.c.hmsg_comm_create(None, 0, 1, int(device_id), C.byref(h))
        if rc != 0:
            raise HmsgError(f"hmsg_comm_create failed ({rc})")
        return cls(h, 0, 1, L)

    @classmethod
    def create(cls, uid: bytes, rank, world, device_id=0, lib_=None):
        L = lib_ or lib()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(uid))
        h = _P()
        rc = L.c.hmsg_comm_create(C.cast(buf, _P), int(rank), int(world), int(device_id), C.byref(h))
        if rc != 0:
            raise HmsgError(f"hmsg_comm_create failed ({rc}): " + (L.c.hmsg_comm_last_error(None) or b"").decode())
        return cls(h, int(rank), int(world), L)

    @staticmethod
    def unique_id(lib_=None) -> bytes:
        L = lib_ or lib()
        buf = (C.c_uint8 * 128)()
        rc = L.c.hmsg_comm_unique_id(C.cast(buf, _P))
        if rc != 0:
            raise HmsgError(f"hmsg_comm_unique_id failed ({rc}): " + (L.c.hmsg_comm_last_error(None) or b"").decode())
        return bytes(buf)

    @classmethod
    def from_torch(cls, device_id=0, lib_=None, group=None):
        import torch.distributed as dist
        rank, world = dist.get_rank(group), dist.get_world_size(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id(lib_)
            except HmsgError:
                box[0] = None                    # (told to the others below: every rank takes the same way out)
        dist.broadcast_object_list(box, src=0, group=group)
        if box[0] is None:
            raise HmsgError("hmsg_comm_unique_id failed on rank 0 (librccl could not be loaded)")
        try:
            c = cls.create(box[0], rank, world, device_id, lib_)
        except HmsgError:
            c = None
        oks = [None] * world
        dist.all_gather_object(oks, c is not None, group=group)
        if not all(oks):
            if c is not None:
                c.close()
            raise HmsgError("hmsg_comm_create failed on ranks %s" % [r for r, ok in enumerate(oks) if not ok])
        return c

    def close(self):
        if getattr(self, "h", None):
            self.L.c.hmsg_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scene:
    """One HMSG scene resident on one GPU (thin handle wrapper over the C ABI)."""

    def __init__(self, cfg: HmsgConfig | None = None, lib_: HmsgLib | None = None, **over):
        self.L = lib_ or lib()
        self.cfg = cfg or self.L.default_config(**over)
        h = _P()
        rc = self.L.c.hmsg_create(C.byref(self.cfg), C.byref(h))
        if rc != 0:
            raise HmsgError(f"hmsg_create failed ({rc})")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.c.hmsg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise HmsgError(f"[{rc}] " + self.L.c.hmsg_last_error(self.h).decode())

    @property
    def HW(self):
        return self.cfg.height * self.cfg.width

    def reset(self):
        self._ck(self.L.c.hmsg_reset(self.h))

    def set_profiling(self, on: bool):
        self._ck(self.L.c.hmsg_set_profiling(self.h, int(on)))

    def profile(self):
        """{kernel name: (launches, total_ms, total algorithmic bytes or FLOP)} from the HIP-event brackets
        recorded since the last reset."""
        out = {}
        for i in range(int(self.L.c.hmsg_profile_count(self.h))):
            name = C.create_string_buffer(64)
            n = C.c_int64()
            ms = C.c_double()
            wk = C.c_double()
            self._ck(self.L.c.hmsg_profile_entry(self.h, i, name, C.byref(n), C.byref(ms), C.byref(wk)))
            out[name.value.decode()] = (int(n.value), float(ms.value), float(wk.value))
        return out

    # ---- build
    def add_frames(self, rgb, depth, pose, K):
        n = int(depth.shape[0])
        self._ck(self.L.c.hmsg_add_frames(self.h, n, _ptr(rgb), _ptr(depth), _ptr(pose), _ptr(K)))

    def finalize_map(self):
        self._ck(self.L.c.hmsg_finalize_map(self.h))

    def map_size(self):
        return int(self.L.c.hmsg_map_size(self.h))

    def num_tie_queries(self):
        return int(self.L.c.hmsg_num_tie_queries(self.h))

    def map_size_unfiltered(self):
        return int(self.L.c.hmsg_map_size_unfiltered(self.h))

    def map_points(self, colors=False):
        V = self.map_size()
        xyz = np.empty((V, 3), np.float64)
        rgb = np.empty((V, 3), np.float64) if colors else None
        self._ck(self.L.c.hmsg_get_map_points(self.h, _ptr(xyz), _ptr(rgb)))
        return (xyz, rgb) if colors else xyz

    def add_frame_features(self, first, masks, f_g, f_masked, f_crop, n_masks=None):
        """masks u8 [n, M, H, W], f_g [n, D], f_masked / f_crop [n, M, D]; n_masks i32 [n] = real masks per frame
        (rows beyond it are padding), None = M for all."""
        n, M = int(f_masked.shape[0]), int(f_masked.shape[1])
        nm = None if n_masks is None else np.ascontiguousarray(n_masks, dtype=np.int32)
        self._ck(self.L.c.hmsg_add_frame_features(self.h, first, n, M, _ptr(masks), _ptr(f_g), _ptr(f_masked), _ptr(f_crop),
                                                  _ptr(nm)))

    def frame_num_masks(self, frame):
        return int(self.L.c.hmsg_get_frame_num_masks(self.h, frame))

    def fuse_frames(self):
        self._ck(self.L.c.hmsg_fuse_frames(self.h))

    def map_feats(self, counter=False):
        V, D = self.map_size(), self.cfg.feat_dim
        f = np.empty((V, D), np.float32)
        c = np.empty((V,), np.float32) if counter else None
        self._ck(self.L.c.hmsg_get_map_feats(self.h, _ptr(f), _ptr(c)))
        return (f, c) if counter else f

    def feature_sums(self):
        V, D = self.map_size(), self.cfg.feat_dim
        s_, c = np.empty((V, D), np.float32), np.empty((V,), np.uint32)
        self._ck(self.L.c.hmsg_get_feature_sums(self.h, _ptr(s_), _ptr(c)))
        return s_, c

    def set_feature_sums(self, sums, counter):
        self._ck(self.L.c.hmsg_set_feature_sums(self.h, _ptr(np.ascontiguousarray(sums, np.float32)),
                                                _ptr(np.ascontiguousarray(counter, np.uint32))))

    def frame_nn(self, frame):
        idx = np.empty((self.cfg.height, self.cfg.width), np.int32)
        self._ck(self.L.c.hmsg_get_frame_nn(self.h, frame, _ptr(idx)))
        return idx

    def frame_fp(self, frame):
        out = np.empty((self.frame_num_masks(frame), self.cfg.feat_dim), np.float32)
        self._ck(self.L.c.hmsg_get_frame_fp(self.h, frame, _ptr(out)))
        return out

    def frame_masks3d(self, frame):
        M = self.frame_num_masks(frame)
        sizes = np.empty((M,), np.int64)
        self._ck(self.L.c.hmsg_get_frame_mask_sizes(self.h, frame, _ptr(sizes)))
        pts = np.empty((int(sizes.sum()), 3), np.float64)
        if pts.shape[0]:
            self._ck(self.L.c.hmsg_get_frame_mask_points(self.h, frame, _ptr(pts)))
        off = np.concatenate([[0], np.cumsum(sizes)])
        return [pts[off[i]:off[i + 1]] for i in range(M)]

    def merge_instances(self):
        self._ck(self.L.c.hmsg_merge_instances(self.h))

    def set_frame_window(self, first_frame):
        self._ck(self.L.c.hmsg_set_frame_window(self.h, int(first_frame)))

    def merge_tree_local(self, total_frames):
        """-> (threshold of the next level, lists at that level, index of this handle's list)"""
        th, n, i = C.c_double(), C.c_int64(), C.c_int64()
        self._ck(self.L.c.hmsg_merge_tree_local(self.h, int(total_frames), C.byref(th), C.byref(n), C.byref(i)))
        return float(th.value), int(n.value), int(i.value)

    def merge_tree_join(self, clouds, th, final_pass):
        """clouds: list of [n, 3] f64 arrays of the partner handle (may be empty)."""
        sizes = np.array([len(c) for c in clouds], np.int64)
        pts = np.ascontiguousarray(np.concatenate(clouds) if len(clouds) and sizes.sum() else np.zeros((0, 3)), np.float64)
        self._ck(self.L.c.hmsg_merge_tree_join(self.h, len(clouds), _ptr(sizes) if len(clouds) else None,
                                               _ptr(pts) if len(pts) else None, float(th), int(bool(final_pass))))

    def merge_tree_join_raw(self, sizes, points, th, final_pass):
        """Same, the partner's clouds as one block: sizes i64 [n] (host), points [sum, 3] f64 -- a numpy array or a torch
        tensor on the host or ON THE DEVICE (e.g. what an RCCL recv just filled)."""
        sizes = np.ascontiguousarray(sizes, np.int64)
        total = int(sizes.sum())
        self._ck(self.L.c.hmsg_merge_tree_join(self.h, len(sizes), _ptr(sizes) if len(sizes) else None,
                                               _ptr(points) if total else None, float(th), int(bool(final_pass))))

    def instance_sizes(self):
        n = int(self.L.c.hmsg_num_instances(self.h))
        sizes = np.empty((n,), np.int64)
        if n:
            self._ck(self.L.c.hmsg_get_instance_sizes(self.h, _ptr(sizes)))
        return sizes

    def instance_points_into(self, out):
        """Instance points into a caller-supplied [sum, 3] f64 buffer (numpy array, or torch tensor on host or device)."""
        self._ck(self.L.c.hmsg_get_instance_points(self.h, _ptr(out)))

    def feature_sums_into(self, sums, counter):
        """Per-voxel feature sums f32 [V, D] and frame counters u32 [V] into caller-supplied buffers (host or device)."""
        self._ck(self.L.c.hmsg_get_feature_sums(self.h, _ptr(sums), _ptr(counter)))

    def set_feature_sums_from(self, sums, counter):
        """Install feature sums / counters from caller-supplied buffers (host or device) without a detour over numpy."""
        self._ck(self.L.c.hmsg_set_feature_sums(self.h, _ptr(sums), _ptr(counter)))

    def instances(self):
        n = int(self.L.c.hmsg_num_instances(self.h))
        sizes = np.empty((n,), np.int64)
        self._ck(self.L.c.hmsg_get_instance_sizes(self.h, _ptr(sizes)))
        pts = np.empty((int(sizes.sum()), 3), np.float64)
        if pts.shape[0]:
            self._ck(self.L.c.hmsg_get_instance_points(self.h, _ptr(pts)))
        off = np.concatenate([[0], np.cumsum(sizes)])
        return [pts[off[i]:off[i + 1]] for i in range(n)]

    def num_instances(self):
        return int(self.L.c.hmsg_num_instances(self.h))

    def instance_boxes(self):
        n = self.num_instances()
        out = np.empty((n, 6), np.float64)
        if n:
            self._ck(self.L.c.hmsg_get_instance_boxes(self.h, _ptr(out)))
        return out

    def instance_room_share(self, room_vertices, radius=0.2):
        """room_vertices: list of [n_r, 2] (x, z) arrays -> share f64 [N, R] (find_intersection_share on device)."""
        R = len(room_vertices)
        off = np.zeros(R + 1, np.int64)
        off[1:] = np.cumsum([len(v) for v in room_vertices])
        verts = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).reshape(-1, 2) for v in room_vertices])
                                     if off[-1] else np.zeros((1, 2)), dtype=np.float64)
        out = np.zeros((self.num_instances(), R), np.float64)
        self._ck(self.L.c.hmsg_instance_room_share(self.h, R, _ptr(off), _ptr(verts), float(radius), _ptr(out)))
        return out

    def build_object_nodes(self, floor_zero, floor_height, room_floor, room_vertices, label_feats=None):
        """A10 on the device (segment_hmsg_objects, graph.py:1582-1736, minus the per-view visibility test): returns the
        node records (numpy structured array: instance, floor, room, counter, label, n_points)."""
        nf, R = len(floor_zero), len(room_vertices)
        fz = np.ascontiguousarray(floor_zero, np.float64)
        fh = np.ascontiguousarray(floor_height, np.float64)
        rf = np.ascontiguousarray(room_floor, np.int32)
        off = np.zeros(R + 1, np.int64)
        off[1:] = np.cumsum([len(v) for v in room_vertices])
        verts = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).reshape(-1, 2) for v in room_vertices])
                                     if off[-1] else np.zeros((1, 2)), dtype=np.float64)
        lf = None if label_feats is None else np.ascontiguousarray(label_feats, np.float32)
        self._ck(self.L.c.hmsg_build_object_nodes(self.h, nf, _ptr(fz), _ptr(fh), R, _ptr(rf), _ptr(off), _ptr(verts),
                                                  0 if lf is None else lf.shape[0], _ptr(lf)))
        return self.nodes()

    def nodes(self, embeddings=False):
        n = int(self.L.c.hmsg_num_nodes(self.h))
        rec = (HmsgNode * max(n, 1))()
        emb = np.empty((n, self.cfg.feat_dim), np.float32) if embeddings else None
        if n:
            self._ck(self.L.c.hmsg_get_nodes(self.h, C.cast(rec, _P), _ptr(emb)))
        out = np.array([(r.instance, r.floor, r.room, r.counter, r.label, r.n_points) for r in rec[:n]],
                       dtype=[("instance", "i4"), ("floor", "i4"), ("room", "i4"), ("counter", "i4"), ("label", "i4"), ("n_points", "i8")])
        return (out, emb) if embeddings else out

    def save_objects(self, directory, records, n_threads=0):
        """Bulk writer of the object level (include/hmsg.h: hmsg_save_objects; Object.save object.py:37-57).  `records`:
        dicts with instance, object_id, room_id, name, view_ids, best_view_id (JSON-encoded here, numbers and clouds by
        the library)."""
        import json
        n = len(records)
        arr = (HmsgObjectRecord * max(n, 1))()
        plain = lambda x: x.item() if isinstance(x, np.generic) else x
        for a, r in zip(arr, records):
            a.instance = int(r["instance"])
            a.file_stem = str(r["object_id"]).encode()
            a.object_id_json = json.dumps(plain(r["object_id"])).encode()
            a.room_id_json = json.dumps(plain(r["room_id"])).encode()
            a.name_json = json.dumps(plain(r["name"])).encode()
            a.view_ids_json = json.dumps([plain(v) for v in r["view_ids"]]).encode()
            a.best_view_id_json = json.dumps(plain(r["best_view_id"])).encode()
        self._ck(self.L.c.hmsg_save_objects(self.h, str(directory).encode(), n, C.cast(arr, _P), int(n_threads)))

    def merge_room_objects(self, clouds, names, overlap_threshold=0.01, radius=0.1):
        """Room.merge_objects (room.py:62-129) for the objects of one room (include/hmsg.h: hmsg_merge_room_objects): `clouds` = the
        objects' point arrays in room.objects order, `names` = their names.  Returns the new room.objects list as groups: [key object,
        objects added to it in the reference's order ...] -- the same-name overlap tests run on the device."""
        n = len(clouds)
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(c) for c in clouds])
        pts = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in clouds]) if off[-1] else np.zeros((1, 3)), np.float64)
        ids = {}
        name_id = np.ascontiguousarray([ids.setdefault(str(v), len(ids)) for v in names], np.int32)
        ng = C.c_int32()
        goff = np.zeros(n + 1, np.int32)
        mem = np.zeros(max(n * (n + 1), 1), np.int32)     # (every object can become a key, and a key's list can hold every other object)
        self._ck(self.L.c.hmsg_merge_room_objects(self.h, n, _ptr(pts), _ptr(off), _ptr(name_id), float(overlap_threshold), float(radius),
                                                  C.byref(ng), _ptr(goff), _ptr(mem), len(mem)))
        return [mem[goff[g]:goff[g + 1]].tolist() for g in range(ng.value)]

    def room_clouds(self, y_lo, y_hi, T, z_levels, room_xz):
        """segment_hmsg_room's room clouds on the device (include/hmsg.h: hmsg_room_clouds): room_xz = list of [n, 2] arrays;
        returns (per room the ascending indices into the floor cloud, size of the floor cloud)."""
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        z = np.ascontiguousarray(z_levels, np.float64).reshape(-1)
        off = np.zeros(len(room_xz) + 1, np.int64)
        off[1:] = np.cumsum([len(r) for r in room_xz])
        xz = np.ascontiguousarray(np.concatenate([np.asarray(r, np.float64).reshape(-1, 2) for r in room_xz]) if off[-1] else np.zeros((0, 2)))
        sizes = np.zeros(len(room_xz), np.int64)
        nf = C.c_int64(0)
        cap = int(len(room_xz)) * max(int(self.map_size()), 1)
        out = np.empty(cap, np.int32)
        self._ck(self.L.c.hmsg_room_clouds(self.h, float(y_lo), float(y_hi), _ptr(T), len(z), _ptr(z), len(room_xz), _ptr(off), _ptr(xz),
                                           _ptr(sizes), _ptr(out), cap, C.byref(nf)))
        o = np.concatenate([[0], np.cumsum(sizes)])
        self._room_clouds_n = len(room_xz)
        return [out[o[r]:o[r + 1]].copy() for r in range(len(room_xz))], int(nf.value)

    def room_camera_distances(self, cam_xz):
        """camera (x, z) -> distance to every room cloud of the last room_clouds call, on the device (include/hmsg.h:
        hmsg_room_camera_distances); [n, R] f64."""
        q = np.ascontiguousarray(cam_xz, np.float64).reshape(-1, 2)
        out = np.empty((len(q), max(self._room_clouds_n, 1)), np.float64)
        self._ck(self.L.c.hmsg_room_camera_distances(self.h, int(self._room_clouds_n), len(q), _ptr(q), _ptr(out)))
        return out[:, : self._room_clouds_n]

    def object_views(self, poses_inv, wh, K, pair_inst, pair_view, min_visible_ratio=0.5, max_depth=10.0):
        """check_object_in_view (utils/graph_utils.py:95-157) for (instance, view) pairs on the device (include/hmsg.h:
        hmsg_object_views): poses_inv [V, 4, 4] world -> camera, wh [V, 2] image width / height, K [3, 3];
        returns (visible bool [P], mean_depth f64 [P])."""
        P_ = np.ascontiguousarray(poses_inv, np.float64).reshape(-1, 16)
        wh = np.ascontiguousarray(wh, np.int32).reshape(-1, 2)
        assert len(wh) == len(P_)
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        pi = np.ascontiguousarray(pair_inst, np.int32).reshape(-1)
        pv = np.ascontiguousarray(pair_view, np.int32).reshape(-1)
        assert len(pi) == len(pv)
        vis = np.zeros(max(len(pi), 1), np.uint8)
        md = np.full(max(len(pi), 1), np.inf)
        self._ck(self.L.c.hmsg_object_views(self.h, len(P_), _ptr(P_), _ptr(wh), _ptr(K), len(pi), _ptr(pi), _ptr(pv),
                                            float(min_visible_ratio), float(max_depth), _ptr(vis), _ptr(md)))
        return vis[:len(pi)].astype(bool), md[:len(pi)]

    def segment_floors(self):
        """A8 behind the C ABI (include/hmsg.h: hmsg_segment_floors) -> list of dicts (y_lo, y_hi, zero_level, height,
        bbox_min, bbox_max, n_points)."""
        n = C.c_int32(0)
        self._ck(self.L.c.hmsg_segment_floors(self.h, None, 0, C.byref(n)))
        rec = np.zeros((max(n.value, 1), 11), np.float64)          # hmsg_floor: 10 doubles + one int64
        self._ck(self.L.c.hmsg_segment_floors(self.h, _ptr(rec), n.value, C.byref(n)))
        return [dict(y_lo=r[0], y_hi=r[1], zero_level=r[2], height=r[3], bbox_min=r[4:7].copy(), bbox_max=r[7:10].copy(),
                     n_points=int(r[10:11].view(np.int64)[0])) for r in rec[:n.value]]

    def segment_rooms(self, y_lo, y_hi, zero_level, height, resolution):
        """N1 on the device (include/hmsg.h: hmsg_segment_rooms) -> (markers i32 [rows, cols], n_rooms, xz_min [2])."""
        rows, cols, nr = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        xz = np.zeros(2, np.float64)
        a = (self.h, float(y_lo), float(y_hi), float(zero_level), float(height), float(resolution))
        self._ck(self.L.c.hmsg_segment_rooms(*a, None, 0, C.byref(rows), C.byref(cols), C.byref(nr), _ptr(xz)))
        m = np.empty((rows.value, cols.value), np.int32)
        self._ck(self.L.c.hmsg_segment_rooms(*a, _ptr(m), m.size, C.byref(rows), C.byref(cols), C.byref(nr), _ptr(xz)))
        return m, int(nr.value), xz

    def allgather_nodes(self, comm: "Comm", n_rooms_local: int):
        """Cross-scene retrieval (include/hmsg.h: hmsg_allgather_nodes): the ranks' node tables all-gathered over RCCL, HBM
        to HBM, into one resident index -> (NodeIndex over the global table, node_off [world + 1], room_off [world + 1])."""
        ix = _P()
        node_off = np.zeros(comm.world + 1, np.int64)
        room_off = np.zeros(comm.world + 1, np.int64)
        self._ck(self.L.c.hmsg_allgather_nodes(self.h, comm.h, int(n_rooms_local), C.byref(ix), _ptr(node_off), _ptr(room_off)))
        return NodeIndex._wrap(self.L, ix, int(node_off[-1]), self.cfg.feat_dim), node_off, room_off

    def allreduce_feature_sums(self, comm: "Comm"):
        """One episode fused in disjoint frame windows (include/hmsg.h: hmsg_allreduce_feature_sums): sums and counters of all
        ranks all-reduced in place over RCCL."""
        self._ck(self.L.c.hmsg_allreduce_feature_sums(self.h, comm.h))

    def merge_tree_sharded(self, comm: "Comm", total_frames: int) -> bool:
        """hierarchical_merge of one episode sharded over the ranks, behind the C ABI (include/hmsg.h: hmsg_merge_tree_sharded): local
        levels, agreement, cross-rank joins over ncclSend / ncclRecv.  True on the rank that holds the episode's instances."""
        holds = C.c_int32()
        self._ck(self.L.c.hmsg_merge_tree_sharded(self.h, comm.h, int(total_frames), C.byref(holds)))
        return bool(holds.value)

    def comm_send(self, comm: "Comm", dev_tensor, dst: int):
        self._ck(self.L.c.hmsg_comm_send(self.h, comm.h, _ptr(dev_tensor), int(dev_tensor.numel() * dev_tensor.element_size()), int(dst)))

    def comm_recv(self, comm: "Comm", dev_tensor, src: int):
        self._ck(self.L.c.hmsg_comm_recv(self.h, comm.h, _ptr(dev_tensor), int(dev_tensor.numel() * dev_tensor.element_size()), int(src)))

    def index_from_nodes(self):
        """Resident retrieval index over the node table, gathered on the device."""
        ix = _P()
        self._ck(self.L.c.hmsg_index_from_nodes(self.h, C.byref(ix)))
        return NodeIndex._wrap(self.L, ix, int(self.L.c.hmsg_num_nodes(self.h)), self.cfg.feat_dim)

    def voxel_down_sample(self, points, voxel_size):
        """Open3D voxel_down_sample of an [n, 3] f64 cloud (canonical ascending voxel order) on the device."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.empty_like(pts)
        n_out = C.c_int64(0)
        self._ck(self.L.c.hmsg_voxel_down_sample(self.h, _ptr(pts), pts.shape[0], float(voxel_size), _ptr(out), C.byref(n_out)))
        return out[: n_out.value].copy()

    def denoise_instances(self, eps=0.05, min_points=10):
        self._ck(self.L.c.hmsg_denoise_instances(self.h, float(eps), int(min_points)))

    def pool_instances(self):
        self._ck(self.L.c.hmsg_pool_instances(self.h))

    def instance_feats(self):
        n = int(self.L.c.hmsg_num_instances(self.h))
        out = np.empty((n, self.cfg.feat_dim), np.float32)
        if n:
            self._ck(self.L.c.hmsg_get_instance_feats(self.h, _ptr(out)))
        return out


class SceneGraph:
    """The graph as one object behind the C ABI (include/hmsg.h: hmsg_build_graph / hmsg_graph_begin + _finish, hmsg_save, hmsg_load,
    hmsg_graph_query) -- floors -> rooms -> views -> objects -> edges held by the library."""

    def __init__(self, L, g, D, scene=None):
        self.L, self.g, self.D, self.scene = L, g, int(D), scene
        self._keep = []

    @staticmethod
    def _params(L, **over):
        p = HmsgGraphParams()
        L.c.hmsg_graph_default_params(C.byref(p))
        for k, v in over.items():
            setattr(p, k, v)
        return p

    @staticmethod
    def _strs(items):
        if items is None:
            return None, None
        enc = [None if s is None else str(s).encode() for s in items]
        arr = (C.c_char_p * max(len(enc), 1))(*enc)
        return arr, enc

    @classmethod
    def begin(cls, scene: "Scene", poses, view_feats, poses_inv=None, img_paths=None, **params):
        """the room level right after hmsg_finalize_map (device stage now, KMeans on host threads); finish() after the pooling"""
        L = scene.L
        P_ = np.ascontiguousarray(np.asarray(poses, np.float64).reshape(-1, 16))
        Pi = None if poses_inv is None else np.ascontiguousarray(np.asarray(poses_inv, np.float64).reshape(-1, 16))
        Fg = np.ascontiguousarray(np.asarray(view_feats, np.float32).reshape(len(P_), -1))
        assert Fg.shape[1] == scene.cfg.feat_dim and (Pi is None or Pi.shape == P_.shape)
        paths, keep = cls._strs(img_paths)
        prm = cls._params(L, **params)
        g = _P()
        scene._ck(L.c.hmsg_graph_begin(scene.h, C.byref(prm), len(P_), _ptr(P_), None if Pi is None else _ptr(Pi), _ptr(Fg),
                                       None if paths is None else C.cast(paths, _P), C.byref(g)))
        return cls(L, g, scene.cfg.feat_dim, scene)

    def finish(self, label_feats=None, label_names=None):
        lf = None if label_feats is None else np.ascontiguousarray(np.asarray(label_feats, np.float32))
        names, keep = self._strs(label_names if lf is not None else None)
        self._ck(self.L.c.hmsg_graph_finish(self.g, 0 if lf is None else len(lf), None if lf is None else _ptr(lf),
                                            None if names is None else C.cast(names, _P)))
        return self

    @classmethod
    def build(cls, scene, poses, view_feats, label_feats=None, label_names=None, poses_inv=None, img_paths=None, **params):
        return cls.begin(scene, poses, view_feats, poses_inv=poses_inv, img_paths=img_paths, **params).finish(label_feats, label_names)

    @classmethod
    def load(cls, directory, device_id=0, lib_: "HmsgLib | None" = None):
        L = lib_ or lib()
        g = _P()
        rc = L.c.hmsg_load(str(directory).encode(), int(device_id), C.byref(g))
        if rc != 0:
            raise HmsgError(f"hmsg_load failed ({rc}) for {directory}")
        return cls(L, g, 0)

    def _ck(self, rc):
        if rc != 0:
            raise HmsgError(self.L.c.hmsg_graph_last_error(self.g).decode() + f" (rc={rc})")

    def close(self):
        if self.g:
            self.L.c.hmsg_graph_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def counts(self):
        c = HmsgGraphCounts()
        self._ck(self.L.c.hmsg_graph_get_counts(self.g, C.byref(c)))
        return {k: getattr(c, k) for k, _ in HmsgGraphCounts._fields_}

    def edges(self):
        n = C.c_int64(0)
        self._ck(self.L.c.hmsg_graph_get_edges(self.g, None, 0, C.byref(n)))
        e = np.zeros((max(n.value, 1), 2), np.int64)
        self._ck(self.L.c.hmsg_graph_get_edges(self.g, _ptr(e), n.value, C.byref(n)))
        return e[: n.value]

    def objects(self):
        n = self.counts()["objects"]
        arr = (HmsgGraphObject * max(n, 1))()
        self._ck(self.L.c.hmsg_graph_get_objects(self.g, C.cast(arr, _P), n))
        return [dict(object_id=a.object_id.decode(), name=a.name.decode(), room=a.room, instance=a.instance, label=a.label,
                     n_views=a.n_views, best_view=a.best_view) for a in arr[:n]]

    def rooms(self):
        n = self.counts()["rooms"]
        arr = (HmsgGraphRoom * max(n, 1))()
        self._ck(self.L.c.hmsg_graph_get_rooms(self.g, C.cast(arr, _P), n))
        return [dict(room_id=a.room_id.decode(), name=a.name.decode(), floor=a.floor, n_vertices=a.n_vertices, n_points=a.n_points,
                     n_embeddings=a.n_embeddings, n_sample_images=a.n_sample_images, n_objects=a.n_objects, n_views=a.n_views)
                for a in arr[:n]]

    def room_vertices(self, room, n=None):
        n = int(n if n is not None else self.rooms()[room]["n_vertices"])
        out = np.zeros((max(n, 1), 2), np.float64)
        self._ck(self.L.c.hmsg_graph_get_room_vertices(self.g, int(room), _ptr(out), out.size))
        return out[:n]

    def room_embeddings(self, room, D=None):
        r = self.rooms()[room]
        D = int(D or self.D)
        out = np.zeros((max(r["n_embeddings"], 1), D), np.float32)
        self._ck(self.L.c.hmsg_graph_get_room_embeddings(self.g, int(room), _ptr(out), out.size))
        return out[: r["n_embeddings"]]

    def to_dict(self):
        import json
        n = C.c_int64(0)
        self._ck(self.L.c.hmsg_graph_to_json(self.g, None, 0, C.byref(n)))
        buf = C.create_string_buffer(n.value)
        self._ck(self.L.c.hmsg_graph_to_json(self.g, buf, n.value, C.byref(n)))
        return json.loads(buf.value.decode())

    def save(self, directory):
        self._ck(self.L.c.hmsg_save(self.g, str(directory).encode()))

    def index(self, room_name_emb=None):
        ix = _P()
        rn = None if room_name_emb is None else np.ascontiguousarray(np.asarray(room_name_emb, np.float64))
        self._ck(self.L.c.hmsg_graph_index(self.g, None if rn is None else _ptr(rn), C.byref(ix)))
        c = self.counts()
        nx = NodeIndex._wrap(self.L, ix, c["objects"], rn.shape[1] if rn is not None else self.D)
        nx._n_rooms = c["rooms"]
        return nx

    def allgather_index(self, comm: "Comm", room_name_emb=None):
        """configs[3] through the graph object (include/hmsg.h: hmsg_graph_allgather_index): the ranks' node tables AND the levels above
        them -- floors -> rooms, room keys, view embeddings, room names -- into one resident index per rank, global ids throughout.
        -> (NodeIndex, node_off, room_off, floor_off), offsets [world + 1]."""
        ix = _P()
        rn = None if room_name_emb is None else np.ascontiguousarray(np.asarray(room_name_emb, np.float64))
        noff, roff, foff = (np.zeros(comm.world + 1, np.int64) for _ in range(3))
        self._ck(self.L.c.hmsg_graph_allgather_index(self.g, comm.h, None if rn is None else _ptr(rn), C.byref(ix), _ptr(noff), _ptr(roff), _ptr(foff)))
        nx = NodeIndex._wrap(self.L, ix, int(noff[-1]), self.D)
        nx._n_rooms = int(roff[-1])
        return nx, noff, roff, foff

    def query(self, T_obj, qid, T_room, floor_id, room_mode, k, use_negatives=True, room_name_emb=None, max_rooms=None):
        """hmsg_graph_query: floor -> room(s) -> objects on the graph's own index (made on the first call)"""
        T_obj = np.ascontiguousarray(T_obj, dtype=np.float32)
        Q, Cn, D = T_obj.shape
        T_room = None if T_room is None else np.ascontiguousarray(T_room, dtype=np.float32)
        qid = np.ascontiguousarray(qid, dtype=np.int32)
        floor_id = np.ascontiguousarray(floor_id, dtype=np.int32)
        room_mode = np.ascontiguousarray(room_mode, dtype=np.int32)
        rn = None if room_name_emb is None else np.ascontiguousarray(np.asarray(room_name_emb, np.float64))
        RM = int(max_rooms or max(self.counts()["rooms"], 10))
        sel, nsel = np.empty((Q, RM), np.int32), np.empty((Q,), np.int32)
        idx, room, score = np.empty((Q, k), np.int32), np.empty((Q, k), np.int32), np.empty((Q, k), np.float64)
        self._ck(self.L.c.hmsg_graph_query(self.g, None if rn is None else _ptr(rn), Q, Cn, _ptr(T_obj), _ptr(qid),
                                           None if T_room is None else _ptr(T_room), _ptr(floor_id), _ptr(room_mode), int(k), int(use_negatives), RM,
                                           _ptr(sel), _ptr(nsel), _ptr(idx), _ptr(room), _ptr(score)))
        return [sel[q, : nsel[q]].tolist() for q in range(Q)], idx, room, score


def points_min_dist_2d(sets, queries, device_id=0, lib_: "HmsgLib | None" = None):
    """out[q][s] = np.min(cdist([queries[q]], sets[s], "euclidean")) on the device (camera -> room assignment of
    compute_room_embeddings, utils/graph_utils.py:244-291)."""
    L = lib_ or lib()
    off = np.zeros(len(sets) + 1, np.int64)
    off[1:] = np.cumsum([len(s_) for s_ in sets])
    pts = np.ascontiguousarray(np.concatenate([np.asarray(s_, np.float64).reshape(-1, 2) for s_ in sets])
                               if off[-1] else np.zeros((1, 2)), dtype=np.float64)
    q = np.ascontiguousarray(queries, dtype=np.float64).reshape(-1, 2)
    out = np.full((len(q), len(sets)), np.inf)
    rc = L.c.hmsg_points_min_dist_2d(device_id, len(sets), _ptr(off), _ptr(pts), len(q), _ptr(q), _ptr(out))
    if rc != 0:
        raise HmsgError(f"hmsg_points_min_dist_2d failed ({rc})")
    return out


def lidar_depth(clouds, poses, intrinsics, width, height, voxel_size=0.02, depth_factor=1000.0, image_scale=1,
                want_state=False, device_id=0, lib_: "HmsgLib | None" = None):
    """LiDAR local maps -> occlusion-aware uint16 depth images on the device (include/hmsg.h: hmsg_lidar_depth; the
    reference's generate_depth.py process_frame per frame).  `clouds`: one [n_f, 3] float64 array per frame; `poses`:
    [F, 3, 4] world -> camera [R | t], or None when the clouds already hold (u, v, z) rows; `intrinsics`: 3x3 K.
    Returns (depth [F, H, W] uint16, stats [F, 4] int64, state per input point or None, device milliseconds)."""
    L = lib_ or lib()
    F = len(clouds)
    off = np.zeros(F + 1, np.int64)
    off[1:] = np.cumsum([len(c) for c in clouds])
    pts = np.ascontiguousarray(np.concatenate([np.asarray(c, np.float64).reshape(-1, 3) for c in clouds])
                               if off[-1] else np.zeros((1, 3)), dtype=np.float64)
    K = np.asarray(intrinsics, np.float64)
    prm = HmsgDepthParams(int(width), int(height), int(image_scale), K[0, 0], K[1, 1], K[0, 2], K[1, 2], float(voxel_size),
                          float(depth_factor))
    T = None
    if poses is not None:
        P = np.asarray(poses, np.float64).reshape(F, 3, 4)
        T = np.ascontiguousarray(np.concatenate([P[:, :, :3].reshape(F, 9), P[:, :, 3]], axis=1))
    depth = np.zeros((F, int(height), int(width)), np.uint16)
    stats = np.zeros((F, 4), np.int64)
    state = np.zeros(max(int(off[-1]), 1), np.uint8) if want_state else None
    ms = C.c_double(0.0)
    rc = L.c.hmsg_lidar_depth(device_id, C.byref(prm), F, _ptr(pts), _ptr(off), _ptr(T) if T is not None else None,
                              _ptr(depth), _ptr(state) if state is not None else None, _ptr(stats), C.byref(ms))
    if rc != 0:
        raise HmsgError(f"hmsg_lidar_depth failed ({rc})")
    return depth, stats, (state[:int(off[-1])] if state is not None else None), ms.value


def write_json_record(path, fields, lib_: "HmsgLib | None" = None):
    """One node record of the on-disk graph through the C ABI (include/hmsg.h: hmsg_write_json) -- byte for byte json.dump's
    text.  fields: (key, value) pairs in the reference's key order; a numpy array of floats (float32 / float64, up to 2-D) or
    of ints is printed by the library, a Python float likewise; everything else (str, None, lists of ids / strings / ints,
    ragged things) goes through json.dumps here and travels as RAW text."""
    import json
    L = lib_ or lib()
    keep, arr = [], (HmsgJsonField * max(len(fields), 1))()
    for k, (key, v) in enumerate(fields):
        f = arr[k]
        f.key = key.encode()
        if isinstance(v, np.ndarray) and v.ndim <= 2 and v.dtype in (np.float64, np.float32) or \
                (isinstance(v, np.ndarray) and v.ndim <= 2 and np.issubdtype(v.dtype, np.integer)):
            if np.issubdtype(v.dtype, np.integer):
                a, kind = np.ascontiguousarray(v, np.int64), 3
            else:
                a, kind = np.ascontiguousarray(v), (1 if v.dtype == np.float64 else 2)
            keep.append(a)
            f.kind, f.ndim = kind, a.ndim
            f.n0 = a.shape[0] if a.ndim >= 1 else 0
            f.n1 = a.shape[1] if a.ndim == 2 else 0
            f.data = a.ctypes.data if a.size else None
        elif isinstance(v, float):
            a = np.array([v], np.float64)
            keep.append(a)
            f.kind, f.ndim, f.n0, f.n1, f.data = 1, 0, 0, 0, a.ctypes.data
        else:
            raw = json.dumps(v).encode()
            keep.append(raw)
            f.kind, f.ndim, f.n0, f.n1 = 0, 0, 0, 0
            f.data = C.cast(C.c_char_p(raw), C.c_void_p)
    rc = L.c.hmsg_write_json(str(path).encode(), len(fields), C.cast(arr, _P))
    if rc != 0:
        raise HmsgError(f"hmsg_write_json failed ({rc}) for {path}")


def write_ply(path, pts, lib_: "HmsgLib | None" = None):
    """A cloud as Open3D's write_point_cloud writes it (include/hmsg.h: hmsg_write_ply)."""
    L = lib_ or lib()
    a = np.ascontiguousarray(np.asarray(pts, np.float64).reshape(-1, 3))
    rc = L.c.hmsg_write_ply(str(path).encode(), _ptr(a) if a.size else None, a.shape[0])
    if rc != 0:
        raise HmsgError(f"hmsg_write_ply failed ({rc}) for {path}")


def read_json_numbers(path, key, lib_: "HmsgLib | None" = None):
    """The numbers under `key` of a saved node record, flattened (include/hmsg.h: hmsg_read_json_numbers)."""
    L = lib_ or lib()
    n = C.c_int64(0)
    rc = L.c.hmsg_read_json_numbers(str(path).encode(), key.encode(), None, 0, C.byref(n))
    if rc != 0:
        raise HmsgError(f"hmsg_read_json_numbers failed ({rc}) for {key} of {path}")
    out = np.empty(max(n.value, 1), np.float64)
    rc = L.c.hmsg_read_json_numbers(str(path).encode(), key.encode(), _ptr(out), n.value, C.byref(n))
    if rc != 0:
        raise HmsgError(f"hmsg_read_json_numbers failed ({rc}) for {key} of {path}")
    return out[: n.value]


def assign_cameras_to_rooms(dist, cam_height, y_min, y_max, lib_: "HmsgLib | None" = None):
    """compute_room_embeddings' camera -> room step (include/hmsg.h: hmsg_assign_cameras_to_rooms).  dist [n_cams, n_rooms].
    Returns (room_of_cam [n_cams] int32, per-room image id lists)."""
    L = lib_ or lib()
    d = np.ascontiguousarray(np.asarray(dist, np.float64))
    n_cams, n_rooms = (int(d.shape[0]), int(d.shape[1])) if d.ndim == 2 else (0, 0)
    h = np.ascontiguousarray(np.asarray(cam_height, np.float64).reshape(-1))
    room_of = np.full(max(n_cams, 1), -1, np.int32)
    off = np.zeros(n_rooms + 1, np.int64)
    imgs = np.zeros(max(n_cams + n_rooms, 1), np.int32)
    rc = L.c.hmsg_assign_cameras_to_rooms(_ptr(d) if d.size else None, n_cams, n_rooms, _ptr(h) if h.size else None, float(y_min),
                                          float(y_max), _ptr(room_of), _ptr(off), _ptr(imgs))
    if rc != 0:
        raise HmsgError(f"hmsg_assign_cameras_to_rooms failed ({rc})")
    return room_of[:n_cams], [imgs[off[r]:off[r + 1]].tolist() for r in range(n_rooms)]


def kmeans(X, n_clusters, n_init=5, max_iter=100, seed=0, lib_: "HmsgLib | None" = None):
    """KMeans(n_clusters, n_init, max_iter, random_state=seed).fit(X) behind the C ABI (include/hmsg.h: hmsg_kmeans; a restatement
    of scikit-learn's Lloyd KMeans for hosts without it).  Returns (labels i32 [n], centers f32 [k, D], inertia, n_iter)."""
    L = lib_ or lib()
    X = np.ascontiguousarray(np.asarray(X, np.float32))
    n, D = X.shape
    labels = np.empty(n, np.int32)
    centers = np.empty((int(n_clusters), D), np.float32)
    inertia, n_iter = C.c_float(0), C.c_int32(0)
    rc = L.c.hmsg_kmeans(_ptr(X), n, D, int(n_clusters), int(n_init), int(max_iter), int(seed), _ptr(labels), _ptr(centers),
                         C.byref(inertia), C.byref(n_iter))
    if rc != 0:
        raise HmsgError(f"hmsg_kmeans failed ({rc})")
    return labels, centers, float(inertia.value), int(n_iter.value)


def pick_representative_views(embs, labels, centers, lib_: "HmsgLib | None" = None):
    """Per KMeans label present (ascending), the member closest (dot product) to its centre (include/hmsg.h:
    hmsg_pick_representative_views).  Returns indices into the room's image list."""
    L = lib_ or lib()
    e = np.ascontiguousarray(np.asarray(embs, np.float32))
    c = np.ascontiguousarray(np.asarray(centers, np.float32))
    lab = np.ascontiguousarray(np.asarray(labels, np.int32).reshape(-1))
    out = np.zeros(max(c.shape[0], 1), np.int32)
    n = C.c_int32(0)
    rc = L.c.hmsg_pick_representative_views(_ptr(e) if e.size else None, e.shape[0], e.shape[1], _ptr(lab) if lab.size else None,
                                            _ptr(c) if c.size else None, c.shape[0], _ptr(out), C.byref(n))
    if rc != 0:
        raise HmsgError(f"hmsg_pick_representative_views failed ({rc})")
    return out[: n.value].tolist()


def graph_edges(n_floors, room_floor, obj_room, view_room, view_objs, lib_: "HmsgLib | None" = None):
    """create_graph_new as an edge list of node ids (0 building, floors, rooms, objects, views; include/hmsg.h:
    hmsg_graph_edges).  view_objs: per view, the positions of its objects.  Returns int64 [n_edges, 2]."""
    L = lib_ or lib()
    rf = np.ascontiguousarray(np.asarray(room_floor, np.int32).reshape(-1))
    orm = np.ascontiguousarray(np.asarray(obj_room, np.int32).reshape(-1))
    vr = np.ascontiguousarray(np.asarray(view_room, np.int32).reshape(-1))
    off = np.zeros(len(vr) + 1, np.int64)
    off[1:] = np.cumsum([len(v) for v in view_objs]) if len(vr) else 0
    vo = np.ascontiguousarray(np.asarray([o for v in view_objs for o in v], np.int32))
    cap = 1 + int(n_floors) + len(rf) + len(orm) + len(vr) + len(vo)
    edges = np.zeros((max(cap, 1), 2), np.int64)
    n = C.c_int64(0)
    rc = L.c.hmsg_graph_edges(int(n_floors), len(rf), _ptr(rf) if rf.size else None, len(orm), _ptr(orm) if orm.size else None, len(vr),
                              _ptr(vr) if vr.size else None, _ptr(off), _ptr(vo) if vo.size else None, _ptr(edges), cap, C.byref(n))
    if rc != 0:
        raise HmsgError(f"hmsg_graph_edges failed ({rc})")
    return edges[: n.value]


def crop_all_bounding_boxs(image, masks, bbox_margin=0, size=512, plain=True, masked=True, device_id=0,
                           lib_: "HmsgLib | None" = None):
    """Both crop sets of one frame on the device (include/hmsg.h: hmsg_crop_resize_batch; utils/sam_utils.py:119-147 as
    called by sam_clip_feats_extractor.py:148-151).  `masks`: SAM records (dicts with "segmentation" and "bbox" XYWH).
    Returns (plain [M, S, S, 3] uint8 or None, masked [M, S, S, 3] uint8 or None)."""
    L = lib_ or lib()
    image = np.ascontiguousarray(image, dtype=np.uint8)
    H, W = image.shape[:2]
    M = len(masks)
    bbox = np.ascontiguousarray([m["bbox"] for m in masks], dtype=np.float64).reshape(M, 4)
    segs = np.ascontiguousarray(np.stack([m["segmentation"] for m in masks]).astype(np.uint8)) if (masked and M) else None
    o_plain = np.zeros((M, size, size, 3), np.uint8) if plain else None
    o_masked = np.zeros((M, size, size, 3), np.uint8) if masked else None
    rc = L.c.hmsg_crop_resize_batch(device_id, H, W, _ptr(image), M, _ptr(segs), _ptr(bbox), float(bbox_margin), int(size),
                                    _ptr(o_plain), _ptr(o_masked), None)
    if rc != 0:
        raise HmsgError(f"hmsg_crop_resize_batch failed ({rc})")
    return o_plain, o_masked


class NodeIndex:
    """Resident node-embedding table for retrieval (graph.py:3056-3162)."""

    def __init__(self, emb: np.ndarray, room_of_node: np.ndarray, device_id=0, lib_: HmsgLib | None = None):
        self.L = lib_ or lib()
        emb = np.ascontiguousarray(emb)
        assert emb.dtype in (np.float32, np.float64)
        room_of_node = np.ascontiguousarray(room_of_node, dtype=np.int32)
        self.N, self.D = emb.shape
        ix = _P()
        rc = self.L.c.hmsg_index_create(device_id, self.D, self.N, _ptr(emb), int(emb.dtype == np.float64),
                                        _ptr(room_of_node), C.byref(ix))
        if rc != 0:
            raise HmsgError(f"hmsg_index_create failed ({rc})")
        self.ix = ix

    @classmethod
    def load_objects(cls, directory, stems, room_of_node, device_id=0, n_threads=0, lib_: "HmsgLib | None" = None):
        """The object table of a saved graph (objects/<stem>.json, "embedding" arrays as float64) straight into a resident
        index (include/hmsg.h: hmsg_index_load_objects)."""
        L = lib_ or lib()
        n = len(stems)
        arr = (C.c_char_p * max(n, 1))(*[str(s_).encode() for s_ in stems])
        rooms = np.ascontiguousarray(room_of_node, dtype=np.int32)
        ix, d = _P(), C.c_int32(0)
        rc = L.c.hmsg_index_load_objects(device_id, str(directory).encode(), n, C.cast(arr, _P), _ptr(rooms), int(n_threads),
                                         C.byref(ix), C.byref(d))
        if rc != 0:
            raise HmsgError(f"hmsg_index_load_objects failed ({rc})")
        return cls._wrap(L, ix, n, int(d.value))

    @classmethod
    def _wrap(cls, L, ix, n, d):
        self = cls.__new__(cls)
        self.L, self.ix, self.N, self.D = L, ix, n, d
        return self

    def close(self):
        if getattr(self, "ix", None):
            self.L.c.hmsg_index_destroy(self.ix)
            self.ix = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise HmsgError(f"[{rc}] " + self.L.c.hmsg_index_last_error(self.ix).decode())

    def set_profiling(self, on=True):
        self._ck(self.L.c.hmsg_index_set_profiling(self.ix, int(on)))

    def profile(self):
        """(launches, total ms, total FLOP) of the float64 MFMA similarity GEMM."""
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        self._ck(self.L.c.hmsg_index_profile(self.ix, C.byref(n), C.byref(ms), C.byref(fl)))
        return int(n.value), float(ms.value), float(fl.value)

    def set_hierarchy(self, floor_rooms, room_name_emb, room_view_embs, room_keys):
        """floor_rooms: per floor the room ids in floors[f].rooms order; room_name_emb f64 [R, D] (or None);
        room_view_embs: per room an [n_v, D] array (room.embeddings); room_keys: per room int(room_id.split("_")[-1])."""
        R = len(room_view_embs)
        foff = np.zeros(len(floor_rooms) + 1, np.int32)
        foff[1:] = np.cumsum([len(r) for r in floor_rooms])
        fr = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int32) for r in floor_rooms]) if foff[-1] else np.zeros(0, np.int32), np.int32)
        voff = np.zeros(R + 1, np.int64)
        voff[1:] = np.cumsum([len(v) for v in room_view_embs])
        views = np.ascontiguousarray(np.concatenate([np.asarray(v, np.float64).reshape(-1, self.D) for v in room_view_embs])
                                     if voff[-1] else np.zeros((0, self.D)), np.float64)
        names = None if room_name_emb is None else np.ascontiguousarray(room_name_emb, np.float64)
        keys = np.ascontiguousarray(room_keys, np.int32)
        self._ck(self.L.c.hmsg_index_set_hierarchy(self.ix, R, len(floor_rooms), _ptr(foff), _ptr(fr), _ptr(names), _ptr(voff),
                                                   _ptr(views), _ptr(keys)))
        self._n_rooms = R

    def query_hier(self, T_obj, qid, T_room, floor_id, room_mode, k, use_negatives=True, max_rooms=None):
        """floor -> room(s) -> objects on the device (include/hmsg.h: hmsg_query_hier).  Returns (rooms per query as
        query_hmsg_room reports them, idx [Q, k], room [Q, k] (global ids), score [Q, k])."""
        # (text rows: numpy arrays, or float32 torch tensors -- on the device they are used where they are: hmsg.h, "host or device pointers")
        if not hasattr(T_obj, "data_ptr"):
            T_obj = np.ascontiguousarray(T_obj, dtype=np.float32)
        Q, Cn, D = T_obj.shape
        if T_room is not None and not hasattr(T_room, "data_ptr"):
            T_room = np.ascontiguousarray(T_room, dtype=np.float32)
        qid = np.ascontiguousarray(qid, dtype=np.int32)
        floor_id = np.ascontiguousarray(floor_id, dtype=np.int32)
        room_mode = np.ascontiguousarray(room_mode, dtype=np.int32)
        RM = int(max_rooms or max(self._n_rooms, 10))
        sel = np.empty((Q, RM), np.int32)
        nsel = np.empty((Q,), np.int32)
        idx = np.empty((Q, k), np.int32)
        room = np.empty((Q, k), np.int32)
        score = np.empty((Q, k), np.float64)
        self._ck(self.L.c.hmsg_query_hier(self.ix, Q, Cn, _ptr(T_obj), _ptr(qid), _ptr(T_room), _ptr(floor_id), _ptr(room_mode), k,
                                          int(use_negatives), RM, _ptr(sel), _ptr(nsel), _ptr(idx), _ptr(room), _ptr(score)))
        return [sel[q, : nsel[q]].tolist() for q in range(Q)], idx, room, score

    def query_objects(self, T, qid, room_lists, k, use_negatives=True):
        T = np.ascontiguousarray(T, dtype=np.float32)
        Q, Cn, D = T.shape
        qid = np.ascontiguousarray(qid, dtype=np.int32)
        off = np.zeros(Q + 1, np.int32)
        off[1:] = np.cumsum([len(r) for r in room_lists])
        rooms = np.ascontiguousarray(np.concatenate([np.asarray(r, np.int32) for r in room_lists]) if off[-1] else
                                     np.zeros(0, np.int32), dtype=np.int32)
        idx = np.empty((Q, k), np.int32)
        room = np.empty((Q, k), np.int32)
        score = np.empty((Q, k), np.float64)
        self._ck(self.L.c.hmsg_query_objects(self.ix, Q, Cn, _ptr(T), _ptr(qid), _ptr(off), _ptr(rooms), k,
                                             int(use_negatives), _ptr(idx), _ptr(room), _ptr(score)))
        return idx, room, score

    def similarity(self, T):
        T = np.ascontiguousarray(T, dtype=np.float32)
        S = np.empty((T.shape[0], self.N), np.float64)
        self._ck(self.L.c.hmsg_similarity(self.ix, T.shape[0], _ptr(T), _ptr(S)))
        return S
