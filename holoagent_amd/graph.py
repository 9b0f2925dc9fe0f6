"""Host-side mirror of the reference's `Graph` (fsr_vln/memory/hmsg/graph/graph.py:77) on top of the C ABI.

Same public surface as the reference for the mapping / retrieval path -- method names, argument meaning,
return shapes, `print` + `return None` error convention -- so the reference apps
(application/.../semantic_scene_reconstruction.py:109-127, visualize_query_graph_icra_*.py:179-265,
nav_agent/.../goal_pose_publisher.py:75-221) run with a one-line import change.

What runs where: everything between the encoders' outputs and the pooled instance embeddings, plus
retrieval, runs in libhmsg (HIP).  This file only orchestrates, assembles the node objects (A8 floors,
A10 object->room/view, A11 node/edge records + on-disk layout) and parses query triples.  Out of scope
(SURVEY section 2): GPT/LLM parsing and slow reasoning, navigation graph, room segmentation by OpenCV
watershed (rooms are an input: `set_rooms`), visualisation.

Collaborators (the same the reference constructs in Graph.__init__, graph.py:98-219), injected so that no
encoder package is needed to import this module:
    dataset[i] -> (rgb PIL/array, depth u16 PIL/array, pose 4x4 f64, _, K 3x3)   (horizon.py:217-268)
    encoders.extract(rgb_array) -> dict(masks bool [M,H,W], f_g [1,D], f_masked [M,D], f_crop [M,D])
        = the encoder half of extract_feats_per_pixel (sam_clip_feats_extractor.py:117-158)
    encoders.encode_text(list[str]) -> float32 [n, D] unit rows  (clip_utils.py:143-162)
"""
from __future__ import annotations

import json
import os
import sys
from typing import List, Sequence

import numpy as np
from scipy.ndimage import gaussian_filter1d
from scipy.signal import find_peaks
from scipy.spatial import cKDTree

from ._lib import HmsgLib, NodeIndex, Scene, lib as _default_lib

CLIP_DIM = {"ViT-B-32": 512, "ViT-L-14": 768, "ViT-H-14": 1024}     # utils/constants.py:3-7
TEXT_TEMPLATES = ["{}", "a photo of {} in the scene."]                 # clip_utils.py:262-265


def _get(cfg, path, default=None):
    cur = cfg
    for k in path.split("."):
        if cur is None:
            return default
        if isinstance(cur, dict):
            cur = cur.get(k, None)
        else:
            cur = getattr(cur, k, None)
    return default if cur is None else cur


def _match_size(rgb, depth):
    """graph.py:342-343 / 378-379: a PIL rgb image whose size differs from the depth image's is resized to it (the
    iPhone loader delivers 1920x1440 colour with 256x192 depth); arrays pass through."""
    if hasattr(rgb, "mode") and hasattr(depth, "mode") and rgb.size != depth.size:
        return rgb.resize(depth.size)
    if not hasattr(rgb, "mode"):
        a, d = np.asarray(rgb), np.asarray(depth)
        assert a.shape[:2] == d.shape[:2], "rgb and depth arrays must have the same height and width"
    return rgb


class _Pcd:
    """Minimal stand-in for the o3d PointCloud attributes callers read (.points, get_center())."""

    def __init__(self, pts=None):
        self.points = np.zeros((0, 3)) if pts is None else np.asarray(pts, dtype=np.float64)

    def get_center(self):
        return self.points.mean(axis=0)

    def is_empty(self):
        return len(self.points) == 0


class _LazyFn:
    """A cloud that stays in HBM until somebody reads `.points` (the map of a resident scene, a storey's slab of it)."""

    def __init__(self, fetch, n=None):
        self._fetch, self._pts, self._n = fetch, None, n

    @property
    def points(self):
        if self._pts is None:
            self._pts = np.asarray(self._fetch(), dtype=np.float64)
        return self._pts

    def get_center(self):
        return self.points.mean(axis=0)

    def is_empty(self):
        return (len(self.points) if self._n is None else self._n) == 0


class _InstanceStore:
    """Instance clouds stay in HBM; the first access downloads all of them once."""

    def __init__(self, scene):
        self.scene = scene
        self.sizes = np.empty((scene.num_instances(),), np.int64)
        if len(self.sizes):
            scene._ck(scene.L.c.hmsg_get_instance_sizes(scene.h, self.sizes.ctypes.data_as(__import__("ctypes").c_void_p)))
        self._clouds = None

    def get(self, i):
        if self._clouds is None:
            self._clouds = self.scene.instances()
        return self._clouds[i]


class _LazyPcd:
    def __init__(self, store, i):
        self._store, self._i = store, i

    @property
    def points(self):
        return self._store.get(self._i)

    def get_center(self):
        return self.points.mean(axis=0)

    def is_empty(self):
        return int(self._store.sizes[self._i]) == 0


def _write_ply(path, pts):
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    with open(path, "wb") as f:
        # (header as Open3D 0.18's write_point_cloud emits it for a cloud without colours / normals, comment line included)
        f.write(("ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex %d\nproperty double x\n"
                 "property double y\nproperty double z\nend_header\n" % len(pts)).encode())
        f.write(pts.astype("<f8").tobytes())


def _read_ply(path):
    with open(path, "rb") as f:
        n, props = 0, []
        while True:
            line = f.readline().decode().strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property"):
                props.append(line.split()[1:])
            elif line == "end_header":
                break
        dt = np.dtype([(p[1], {"double": "<f8", "float": "<f4", "uchar": "u1"}[p[0]]) for p in props])
        arr = np.frombuffer(f.read(n * dt.itemsize), dtype=dt, count=n)
    return np.stack([arr["x"], arr["y"], arr["z"]], axis=1).astype(np.float64)


def _box_points(pts):
    """Open3D AxisAlignedBoundingBox::GetBoxPoints order of the AABB of `pts`."""
    mn, mx = pts.min(0), pts.max(0)
    ex = mx - mn
    return np.array([mn, mn + [ex[0], 0, 0], mn + [0, ex[1], 0], mn + [0, 0, ex[2]], mx,
                     mn + [0, ex[1], ex[2]], mn + [ex[0], 0, ex[2]], mn + [ex[0], ex[1], 0]])


def find_overlapping_ratio_faiss(p1, p2, radius=0.02):
    """utils/graph_utils.py:620-664 (faiss IndexFlatL2, k=1): fraction of points whose exact float32 nearest
    neighbour in the other cloud is closer than radius (float32 `(dx*dx + dy*dy) + dz*dz < radius**2`), max over
    both directions.  Host version for the few-hundred-point object clouds of Room.merge_objects."""
    p1, p2 = np.asarray(getattr(p1, "points", p1)), np.asarray(getattr(p2, "points", p2))
    if p1.shape[0] == 0 or p2.shape[0] == 0:
        return 0
    a, b = p1.astype(np.float32), p2.astype(np.float32)

    def nn_d2(q, base):
        _, nn = cKDTree(base.astype(np.float64)).query(q.astype(np.float64), k=1, workers=-1)
        d = q - base[nn]
        return (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
    r2 = np.float32(radius ** 2)
    return np.max([np.sum(nn_d2(a, b) < r2) / a.shape[0], np.sum(nn_d2(b, a) < r2) / b.shape[0]])


def _num(v):
    """a scalar for write_json_record: numpy / Python numbers as Python float (json.dump prints float.__repr__), None stays"""
    return float(v) if isinstance(v, (float, np.floating)) else v


def _num2d(v):
    """`np.asarray(v).tolist()` of a coordinate table as the float64 array the library prints (anything else: as a list)"""
    a = np.asarray(v)
    return np.ascontiguousarray(a, np.float64) if a.ndim == 2 and a.dtype.kind == "f" else a.tolist()


def _rows(rows):
    """`[np.asarray(e).tolist() for e in rows]`: equally long float rows go to the library as ONE 2-D array (float32 rows are
    printed as the double of each float32, which is what tolist() hands json.dump); ragged / empty lists stay lists"""
    rows = [np.asarray(e) for e in rows]
    if rows and all(r.ndim == 1 and r.dtype == rows[0].dtype and r.shape == rows[0].shape and r.dtype in (np.float32, np.float64) for r in rows):
        return np.ascontiguousarray(np.stack(rows))
    return [r.tolist() for r in rows]


class Floor:   # graph/floor.py:10-67
    def __init__(self, floor_id, name=None):
        self.floor_id, self.name = floor_id, name
        self.rooms, self.pcd, self.vertices = [], _Pcd(), np.zeros((8, 3))
        self.floor_height = self.floor_zero_level = None

    def add_room(self, room):
        self.rooms.append(room)

    def save(self, path, lib=None):
        """floor.py:37-52.  lib: write through the C ABI (hmsg_write_ply / hmsg_write_json: the same bytes, numbers printed by
        the library) -- what Graph.save_hmsg_graph does, and what a C / C++ host calls directly."""
        if lib is not None:
            from ._lib import write_json_record, write_ply
            write_ply(os.path.join(path, str(self.floor_id) + ".ply"), self.pcd.points, lib)
            write_json_record(os.path.join(path, str(self.floor_id) + ".json"),
                              [("floor_id", self.floor_id), ("name", self.name), ("rooms", [r.room_id for r in self.rooms]),
                               ("vertices", _num2d(self.vertices)), ("floor_height", _num(self.floor_height)),
                               ("floor_zero_level", _num(self.floor_zero_level))], lib)
            return
        _write_ply(os.path.join(path, str(self.floor_id) + ".ply"), self.pcd.points)
        meta = dict(floor_id=self.floor_id, name=self.name, rooms=[r.room_id for r in self.rooms],
                    vertices=np.asarray(self.vertices).tolist(), floor_height=self.floor_height,
                    floor_zero_level=self.floor_zero_level)
        json.dump(meta, open(os.path.join(path, str(self.floor_id) + ".json"), "w"))

    def load(self, path):
        self.pcd = _Pcd(_read_ply(os.path.join(path, str(self.floor_id) + ".ply")))
        m = json.load(open(os.path.join(path, str(self.floor_id) + ".json")))
        self.name, self.rooms, self.vertices = m["name"], m["rooms"], np.asarray(m["vertices"])
        self.floor_height, self.floor_zero_level = m["floor_height"], m["floor_zero_level"]


class Room:    # graph/room.py:15-60, 309-374
    def __init__(self, room_id, floor_id, name=None):
        self.room_id, self.floor_id, self.name = room_id, floor_id, name
        self.objects, self.views, self.pcd, self.vertices = [], [], _Pcd(), np.zeros((0, 2))
        self.embeddings, self.clip_embeddings, self.represent_images, self.sample_images = [], [], [], []
        self.room_height = self.room_zero_level = None
        self.object_counter = 0

    def add_object(self, obj):
        self.objects.append(obj)

    def merge_groups(self, overlap_threshold=0.01, radius=0.1):
        """room.py:62-106, host form: the new room.objects list as groups [key object, objects added to it ...] -- numpy / cKDTree
        overlap tests, Python's own dict and set (the library's hmsg_merge_room_objects restates both and is compared with this)."""
        n = len(self.objects)
        scores = np.zeros((n, n))
        for i in range(n):
            for j in range(i + 1, n):
                if self.objects[i].name == self.objects[j].name:
                    ov = find_overlapping_ratio_faiss(self.objects[i].pcd, self.objects[j].pcd, radius)
                    if ov > overlap_threshold:
                        scores[i, j] = scores[j, i] = ov
        groups: dict = {}
        merging = []
        for i, j in zip(*np.where(scores > 0)):
            i, j = int(i), int(j)
            merging.extend([i, j])
            if i not in groups and j not in groups:
                groups.setdefault(i, []).append(j)
            elif i in groups:
                groups[i].append(j)
            elif j in groups:
                groups[j].append(i)
        merging = set(merging)
        for idx in range(n):
            if idx not in merging:
                groups.setdefault(idx, []).append(idx)
        out = []
        for i, js in groups.items():
            js = list(set(js))
            out.append([i] if (len(js) == 1 and js[0] == i) else [i] + js)
        return out

    def merge_objects(self, overlap_threshold=0.01, radius=0.1, scene=None):
        """room.py:62-129: fuse same-name objects of the room whose clouds overlap, then re-number the objects.
        Follows the reference step by step (including what its dictionary bookkeeping does with chains).  scene: the overlap
        tests and the bookkeeping behind the C ABI (hmsg_merge_room_objects: tests on the device); None: numpy on the host."""
        if scene is not None:
            groups = scene.merge_room_objects([np.asarray(o.pcd.points).reshape(-1, 3) for o in self.objects], [o.name for o in self.objects],
                                              overlap_threshold, radius)
        else:
            groups = self.merge_groups(overlap_threshold, radius)
        new_objects = []
        for counter, g in enumerate(groups):
            obj = self.objects[g[0]]
            for jj in g[1:]:
                obj = obj + self.objects[jj]                   # (Object.__add__: an empty side yields the other object)
                obj.object_id = self.room_id + "_" + str(counter)
            obj.object_id = self.room_id + "_" + str(counter)
            new_objects.append(obj)
        self.objects = new_objects

    def infer_room_type_from_view_embedding(self, default_room_types, text_feats):
        """room.py:131-172: per view arg-max over room-type text features, majority vote, smallest type id on ties."""
        if len(self.embeddings) == 0:
            return "unknown room type"                                     # room.py:153-155 (the name is left as it was)
        votes = np.argmax(np.dot(np.stack(self.embeddings), np.asarray(text_feats).T), axis=1)
        cnt = np.bincount(votes, minlength=len(default_room_types))
        self.name = default_room_types[int(np.argmax(cnt))]
        return self.name

    def save(self, path, lib=None):
        """room.py:309-337 (lib: through the C ABI, see Floor.save)"""
        if lib is not None:
            from ._lib import write_json_record, write_ply
            write_ply(os.path.join(path, str(self.room_id) + ".ply"), self.pcd.points, lib)
            write_json_record(os.path.join(path, str(self.room_id) + ".json"),
                              [("room_id", self.room_id), ("name", self.name), ("floor_id", self.floor_id),
                               ("objects", [o.object_id for o in self.objects]),
                               ("views", [v.view_id if hasattr(v, "view_id") else v for v in self.views]),
                               ("vertices", _num2d(self.vertices)), ("room_height", _num(self.room_height)),
                               ("room_zero_level", _num(self.room_zero_level)), ("embeddings", _rows(self.embeddings)),
                               ("represent_images", self.represent_images), ("sample_images", self.sample_images),
                               ("clip_embeddings", _rows(self.clip_embeddings))], lib)
            return
        _write_ply(os.path.join(path, str(self.room_id) + ".ply"), self.pcd.points)
        meta = dict(room_id=self.room_id, name=self.name, floor_id=self.floor_id,
                    objects=[o.object_id for o in self.objects],
                    views=[v.view_id if hasattr(v, "view_id") else v for v in self.views],
                    vertices=np.asarray(self.vertices).tolist(), room_height=self.room_height,
                    room_zero_level=self.room_zero_level, embeddings=[np.asarray(e).tolist() for e in self.embeddings],
                    represent_images=self.represent_images, sample_images=self.sample_images,
                    clip_embeddings=[np.asarray(e).tolist() for e in self.clip_embeddings])
        json.dump(meta, open(os.path.join(path, str(self.room_id) + ".json"), "w"))

    def load_new(self, path):
        self.pcd = _Pcd(_read_ply(os.path.join(path, str(self.room_id) + ".ply")))
        m = json.load(open(os.path.join(path, str(self.room_id) + ".json")))
        self.name, self.floor_id, self.vertices = m["name"], m["floor_id"], np.asarray(m["vertices"])
        self.room_height, self.room_zero_level = m["room_height"], m["room_zero_level"]
        self.embeddings = [np.asarray(i) for i in m["embeddings"]]
        self.represent_images, self.sample_images = m["represent_images"], m["sample_images"]
        self.clip_embeddings = [np.asarray(i) for i in m["clip_embeddings"]]
        self.views = m["views"]


class Object:  # graph/object.py:9-106
    def __init__(self, object_id, room_id, name=None):
        self.object_id, self.room_id, self.name = object_id, room_id, name
        self.pcd, self.vertices, self.embedding = _Pcd(), None, None
        self.view_ids, self.best_view_id = [], None

    def __add__(self, other):
        """object.py:93-103: an empty side yields the other object; otherwise the clouds are concatenated in place,
        vertices become the 8 AABB corners and the embedding the mean of the two."""
        if self.pcd.is_empty():
            return other
        if other.pcd.is_empty():
            return self
        self.pcd = _Pcd(np.concatenate([np.asarray(self.pcd.points).reshape(-1, 3), np.asarray(other.pcd.points).reshape(-1, 3)]))
        self.vertices = _box_points(self.pcd.points)
        self.embedding = np.mean([self.embedding, other.embedding], axis=0)
        return self

    def save(self, path):
        _write_ply(os.path.join(path, str(self.object_id) + ".ply"), self.pcd.points)
        verts = self.vertices if self.vertices is not None else np.asarray(self.pcd.points)[:, [0, 2]]
        meta = dict(object_id=self.object_id, vertices=np.asarray(verts).tolist(), room_id=self.room_id,
                    name=self.name, embedding=self.embedding.tolist() if self.embedding is not None else "",
                    view_ids=self.view_ids, best_view_id=self.best_view_id)
        json.dump(meta, open(os.path.join(path, str(self.object_id) + ".json"), "w"))

    def load_new(self, path):
        self.pcd = _Pcd(_read_ply(os.path.join(path, str(self.object_id) + ".ply")))
        m = json.load(open(os.path.join(path, str(self.object_id) + ".json")))
        self.vertices, self.room_id, self.name = np.asarray(m["vertices"]), m["room_id"], m["name"]
        self.embedding = np.asarray(m["embedding"]) if m["embedding"] != "" else None   # float64 after load (:88-89)
        self.view_ids, self.best_view_id = m["view_ids"], m["best_view_id"]


class View:    # graph/view.py:36-103
    def __init__(self, view_id, room_id, img_id=None, name=None):
        self.view_id, self.room_id, self.img_id, self.name = view_id, room_id, img_id, name
        self.object_ids, self.text_discription, self.img_path = [], [], None

    def save(self, path, lib=None):
        plain = lambda x: int(x) if isinstance(x, np.integer) else x          # view.py:63-71: numpy ids -> int, text -> str
        meta = dict(view_id=plain(self.view_id), room_id=plain(self.room_id), img_id=plain(self.img_id),
                    object_ids=[plain(x) for x in self.object_ids], img_path=self.img_path,
                    text_discription=[str(x) for x in self.text_discription])
        if lib is not None:                                                   # (no numbers to print: every field travels as text)
            from ._lib import write_json_record
            write_json_record(os.path.join(path, str(self.view_id) + ".json"), list(meta.items()), lib)
            return
        json.dump(meta, open(os.path.join(path, str(self.view_id) + ".json"), "w", encoding="utf-8"))

    def load(self, path):
        m = json.load(open(os.path.join(path, str(self.view_id) + ".json")))
        self.room_id, self.img_id, self.object_ids = m["room_id"], m["img_id"], m["object_ids"]
        self.img_path, self.text_discription = m.get("img_path"), m.get("text_discription", [])


def check_object_in_view(img_w, img_h, K, cam_pose_inv, pts, min_visible_ratio=0.5, max_depth=10.0):
    """utils/graph_utils.py:95-157."""
    if pts.shape[0] == 0:
        return False, np.inf
    cam = (cam_pose_inv @ np.hstack([pts, np.ones((pts.shape[0], 1))]).T).T[:, :3]
    cam = cam[cam[:, 2] > 0]
    if cam.shape[0] == 0:
        return False, np.inf
    px = (K @ cam.T).T
    px = px[:, :2] / px[:, 2:3]
    inside = (px[:, 0] >= 0) & (px[:, 0] < img_w) & (px[:, 1] >= 0) & (px[:, 1] < img_h)
    if not np.any(inside) or np.sum(inside) / pts.shape[0] < min_visible_ratio:
        return False, np.inf
    md = float(np.mean(cam[inside, 2]))
    return (md <= max_depth), md


def find_intersection_share(map_points, obj_points, radius=0.05):
    """utils/graph_utils.py:160-189."""
    if map_points.shape[0] == 0 or obj_points.shape[0] == 0:
        return 0
    _, idx = cKDTree(obj_points).query(map_points, k=1, distance_upper_bound=radius, p=2, workers=-1)
    return int((idx != obj_points.shape[0]).sum()) / obj_points.shape[0]


def camera_room_distances(room_pcds, pose_list, lib=None, device_id=0):
    """The distance table of compute_room_embeddings (graph_utils.py:244-265): camera (x, z) -> nearest (x, z) point of every
    room cloud, [F, R], on the device (hmsg_points_min_dist_2d)."""
    from ._lib import points_min_dist_2d
    flat = [np.stack([np.asarray(getattr(p, "points", p))[:, 0], np.asarray(getattr(p, "points", p))[:, 2]], axis=1)
            for p in room_pcds]
    cam = np.array([[pose[0, 3], pose[2, 3]] for pose in pose_list], dtype=np.float64).reshape(-1, 2)
    return points_min_dist_2d(flat, cam, device_id=device_id, lib_=lib) if len(cam) and len(flat) else np.zeros((len(cam), len(flat)))


def _closest_member(cluster, centre):
    """graph_utils.py:344-346: the member with the largest dot product with its KMeans centre -- float32 np.dot, i.e. BLAS decides
    an exact tie (a two-member cluster: both are equally far from their mean).  hmsg_pick_representative_views accumulates the
    products in float64 and takes the first maximum; tests that compare the two sides byte for byte install that rule here."""
    return int(np.argmax(np.dot(cluster, centre)))


def compute_room_embeddings(room_pcds, pose_list, emb_list, pcd_min, pcd_max, num_views=5, save_path=None, lib=None,
                            device_id=0, dist=None):
    """utils/graph_utils.py:192-356.  Assign every image to the room whose cloud (projected to x/z) is nearest to the
    camera position, provided the camera height lies inside the floor bounds; a room that got no image takes the
    closest of the cameras OUTSIDE the floor bounds (sic, :267-291; image 0 when there is none); per room with at
    least `num_views` images, KMeans(num_views, n_init=5, max_iter=100, random_state=0) over the image embeddings and
    the member closest (dot product) to every centre.  The camera-to-room distances (F x R x room points) are
    computed on the device (camera_room_distances; `dist`: that table, already made); KMeans is scikit-learn's, as in
    the reference.
    Returns (repr_embs_list, repr_img_ids_list, room_id2img_id, room_clip_embeddings_list)."""
    from collections import defaultdict
    from sklearn.cluster import KMeans
    n_rooms = len(room_pcds)
    room_id2img_id = defaultdict(list)
    height = np.array([pose[1, 3] for pose in pose_list], dtype=np.float64)
    if dist is None:
        dist = camera_room_distances(room_pcds, pose_list, lib=lib, device_id=device_id)
    inside = ~((height < pcd_min[1]) | (height > pcd_max[1]))
    for i in range(len(pose_list)):
        if not inside[i]:
            continue
        room_id2img_id[int(np.argmin(dist[i]))].append(i)
    for room_id in range(n_rooms):
        if room_id not in room_id2img_id:
            closest = np.where(inside, np.inf, dist[:, room_id]) if len(pose_list) else np.zeros(0)
            room_id2img_id[room_id].append(int(np.argmin(closest)))
    repr_img_ids_list, repr_embs_list, room_clip_embeddings_list = [], [], []
    # KMeans stays scikit-learn's (the reference's own third-party call, :329-333), run with ONE OpenMP / BLAS thread: a fit
    # over ~10^2 rows is 3x slower with a many-thread pool than with one, and a fit over fewer rows than one Lloyd chunk
    # (256) is the same arithmetic for any thread count.  (The fits are bound by scikit-learn's Python-level work, so host
    # threads do not speed them up; Graph.start_room_level runs this whole stage beside the fusion and the merge fold.)
    clips = {}
    for room_id in range(n_rooms):
        img_ids = room_id2img_id[room_id]
        if len(img_ids):
            clips[room_id] = np.squeeze(np.array([emb_list[i] for i in img_ids]), axis=1)
    todo = [r for r in clips if len(room_id2img_id[r]) >= num_views]
    fits = {}
    if todo:
        try:
            from threadpoolctl import threadpool_limits
        except Exception:
            import contextlib
            threadpool_limits = lambda limits=None: contextlib.nullcontext()
        # (only scikit-learn's own OpenMP pool is limited: limits=1 without user_api also throttles every BLAS pool of the process --
        #  numpy / torch work on other threads -- for as long as the fits run)
        try:
            ctx = threadpool_limits(limits=1, user_api="openmp")
        except TypeError:
            ctx = threadpool_limits(limits=1)
        with ctx:
            fits = {r: KMeans(n_clusters=num_views, max_iter=100, n_init=5, random_state=0).fit(clips[r]) for r in todo}
    for room_id in range(n_rooms):
        img_ids = room_id2img_id[room_id]
        if len(img_ids) == 0:
            repr_img_ids_list.append([])
            repr_embs_list.append([])
            continue
        room_clip = clips[room_id]
        room_clip_embeddings_list.append(room_clip)
        if len(img_ids) < num_views:
            repr_img_ids_list.append(img_ids)
            repr_embs_list.append([emb for emb in room_clip])
            continue
        kmeans = fits[room_id]
        labels, centers = kmeans.labels_, kmeans.cluster_centers_
        repr_img_ids, repr_embs = [], []
        for lab in np.unique(labels):
            ids = np.where(labels == lab)[0]
            cluster = room_clip[ids]
            max_idx = _closest_member(cluster, centers[lab])
            repr_img_ids.append(img_ids[ids[max_idx]])
            repr_embs.append(cluster[max_idx])
        repr_img_ids_list.append(repr_img_ids)
        repr_embs_list.append(repr_embs)
    return repr_embs_list, repr_img_ids_list, room_id2img_id, room_clip_embeddings_list


def parse_hier_instruction(text):
    """Stand-in for the reference's LLM parse (utils/llm_utils parse_hier_query_use_prompt_insentence_parse[_icra], out of
    scope: SURVEY section 2): "<object> in [the] <room> on floor <n>" -> (floor | None, room, object).  Install a real
    parser with `Graph.instruction_parser = callable`."""
    import re
    t = text.strip().rstrip(".")
    floor = None
    m = re.search(r"\s+on\s+(?:the\s+)?floor\s+(\w+)\s*$", t, flags=re.I) or re.search(r"\s+on\s+the\s+(\w+)\s+floor\s*$", t, flags=re.I)
    if m:
        floor, t = m.group(1), t[: m.start()]
    m = re.search(r"\s+in\s+(?:the\s+)?(.+)$", t, flags=re.I)
    room = "unknown"
    if m:
        room, t = m.group(1).strip(), t[: m.start()]
    obj = re.sub(r"^(?:find|go to|navigate to|bring me to|take me to)\s+(?:the\s+|a\s+|an\s+)?", "", t.strip(), flags=re.I)
    return floor, room, obj.strip()


BACKGROUND_LABELS = ["background", "divider", "ledge", "pillar", "tape", "stairs", "door", "doors", "stair", "window", "glass",
                     "railing", "glass doors", "whiteboard", "sliding door", "carpet", "ceiling", "curtain"]   # graph.py:3607-3625


class Graph:
    # str -> (floor_query | None, room_query, object_query); the reference asks an LLM (out of scope), see
    # parse_hier_instruction.  collaborators: cfg -> (dataset, encoders), the part of Graph.__init__ (graph.py:98-219)
    # that builds the dataset reader, SAM and CLIP from the Hydra keys; install the project's own factory to get the
    # reference's `Graph(cfg)` call signature.
    instruction_parser = staticmethod(parse_hier_instruction)
    collaborators = None

    def __init__(self, cfg, dataset=None, encoders=None, lib: HmsgLib | None = None):
        self.cfg = cfg
        self.L = lib or _default_lib()
        if dataset is None and encoders is None and Graph.collaborators is not None:
            dataset, encoders = Graph.collaborators(cfg)
        self.dataset, self.encoders = dataset, encoders
        import networkx as nx
        self.graph = nx.Graph()                                        # graph.py:93
        self.room_id2img_ids = {}
        self._view_feats = []          # per processed frame the global CLIP feature F_g [1, D] (loop B hands it over)
        self.floors: List[Floor] = []
        self.rooms: List[Room] = []
        self.objects: List[Object] = []
        self.views: List[View] = []
        self.mask_pcds, self.mask_feats, self.full_feats_array = [], [], []
        self.full_pcd = _Pcd()
        clip_type = str(_get(cfg, "models.clip.type", "ViT-B/32")).replace("/", "-")
        self.clip_feat_dim = int(_get(cfg, "models.clip.feat_dim", CLIP_DIM.get(clip_type, 512)))
        self.build_mode = _get(cfg, "pipeline") is not None           # graph.py:166-168
        if self.build_mode:
            # every pipeline.* key is accounted for: honoured, the collaborators' business, ignored like the reference ignores it --
            # or refused (an unknown key, a value this path cannot honour): holoagent_amd/config_surface.py
            from .config_surface import check_config
            self.config_report = check_config(cfg)
        self.scene: Scene | None = None
        self._index: NodeIndex | None = None
        self._text_cache = {}
        self.graph_path = _get(cfg, "main.graph_path")
        self._label_feats = None       # (text feats, class names) for identify_object
        self._K, self._poses = None, []

    # ------------------------------------------------------------------ text features
    def get_text_feats_multiple_templates(self, words: Sequence[str]) -> np.ndarray:
        """clip_utils.py:257-349: mean over the 2 templates of unit vectors, NOT re-normalised."""
        miss = [w for w in words if w not in self._text_cache]
        if miss:
            prompts = [t.format(w) for w in miss for t in TEXT_TEMPLATES]
            f = np.asarray(self.encoders.encode_text(prompts), dtype=np.float32)
            f = f.reshape(len(miss), len(TEXT_TEMPLATES), -1).mean(axis=1)
            for w, v in zip(miss, f):
                self._text_cache[w] = v
        return np.stack([self._text_cache[w] for w in words]).astype(np.float32)

    # ------------------------------------------------------------------ build: graph.py:262-491
    def create_feature_map(self, save_path=None):
        if self.dataset is None:
            print("No dataset loaded")
            return
        p = lambda k, d=None: _get(self.cfg, "pipeline." + k, d)
        skip = int(p("skip_frames", 1))
        ids = list(range(0, len(self.dataset), skip))
        first = self.dataset[ids[0]]
        depth0 = np.asarray(first[1])
        H, W = depth0.shape[:2]
        merge = {"sequential": 0, "hierarchical": 1}[str(p("merge_type", "sequential"))]
        self.scene = Scene(lib_=self.L, device_id=int(_get(self.cfg, "main.device_id", 0)), feat_dim=self.clip_feat_dim,
                           height=H, width=W, max_frames=len(ids), max_masks=int(p("max_masks", 256)),
                           voxel_size=float(p("voxel_size", 0.05)), init_overlap_thresh=float(p("init_overlap_thresh", 0.75)),
                           overlap_thresh_factor=float(p("overlap_thresh_factor", 0.025)), iou_thresh=float(p("iou_thresh", 0.05)),
                           clip_masked_weight=float(p("clip_masked_weight", 0.4418)),
                           max_mask_distance=float(p("max_mask_distance", 10000)), merge_type=merge)
        sc = self.scene
        B = 32
        self._poses, self._K = [], None
        for b0 in range(0, len(ids), B):                                  # loop A (graph.py:339-345)
            fr = [self.dataset[i] for i in ids[b0:b0 + B]]
            fr = [(_match_size(f[0], f[1]),) + tuple(f[1:]) for f in fr]    # graph.py:342-343
            rgb = np.ascontiguousarray(np.stack([np.asarray(f[0], dtype=np.uint8)[..., :3] for f in fr]))
            dep = np.ascontiguousarray(np.stack([np.asarray(f[1]).astype(np.uint16) for f in fr]))
            pose = np.ascontiguousarray(np.stack([np.asarray(f[2], dtype=np.float64) for f in fr]))
            self._K = np.ascontiguousarray(np.asarray(fr[0][4], dtype=np.float64))
            self._poses.extend(list(pose))
            sc.add_frames(rgb, dep, pose, self._K)
        sc.finalize_map()
        self.full_pcd = _LazyFn(sc.map_points, sc.map_size())           # stays in HBM until read
        n_done = 0
        D = self.clip_feat_dim
        for b0 in range(0, len(ids), B):                                  # loop B (graph.py:373-411)
            outs = []
            for i in ids[b0:b0 + B]:
                rgb_i = _match_size(self.dataset[i][0], self.dataset[i][1])     # graph.py:378-379
                outs.append(self.encoders.extract(np.asarray(rgb_i)))
            # SAM returns a different number of masks for every frame: rows are padded to the batch maximum and the
            # real counts are handed over with them (sam_clip_feats_extractor.py:167-169 softmaxes over the frame's own)
            n_masks = np.array([np.asarray(o["masks"]).shape[0] for o in outs], np.int32)
            M = max(int(n_masks.max()), 1)

            def pad(a, tail):
                out = np.zeros((M,) + tail, a.dtype)
                out[: a.shape[0]] = a.reshape((a.shape[0],) + tail)
                return out
            masks = np.ascontiguousarray(np.stack([pad(np.asarray(o["masks"]).astype(np.uint8), (H, W)) for o in outs]))
            fg = np.ascontiguousarray(np.stack([np.asarray(o["f_g"], np.float32).reshape(-1) for o in outs]))
            fm = np.ascontiguousarray(np.stack([pad(np.asarray(o["f_masked"], np.float32), (D,)) for o in outs]))
            fc = np.ascontiguousarray(np.stack([pad(np.asarray(o["f_crop"], np.float32), (D,)) for o in outs]))
            sc.add_frame_features(n_done, masks, fg, fm, fc, n_masks)
            self._view_feats.extend(np.asarray(o["f_g"], np.float32).reshape(1, -1) for o in outs)
            n_done += len(outs)
        # the room level (floors, room regions, room clouds, camera -> room table on the device; KMeans views on a host thread)
        # needs only the map and the frames' global features: its device stage runs here, KMeans beside the fusion and the fold
        # (start_room_level; started after hmsg_fuse_frames instead, its launches and the fold's short kernels slow each other down)
        if bool(p("room_level_beside_fusion", True)) and len(self._view_feats) == len(ids):
            try:
                self.start_room_level()
            except Exception as e:                                         # build_hier_multimodal_scene_graph does it in place
                print("room level not started early:", e)
                self.floors, self._room_level = [], None
        sc.fuse_frames()
        self.full_feats_array = sc.map_feats()
        sc.merge_instances()
        sc.pool_instances()
        self.mask_feats = list(sc.instance_feats())
        self.mask_pcds = [_Pcd(x) for x in sc.instances()]
        self._frame_ids = ids
        assert len(self.mask_pcds) == len(self.mask_feats)

    # ------------------------------------------------------------------ A8: graph.py:624-787
    def _voxel_down_sample(self, pts, voxel_size):
        """Open3D voxel_down_sample on the device (a scratch handle when this Graph was not built from a scene)."""
        if self.scene is not None:
            return self.scene.voxel_down_sample(pts, voxel_size)
        from ._lib import Scene
        tmp = Scene(lib_=self.L, height=8, width=8, max_frames=1, max_masks=1, feat_dim=8)
        try:
            return tmp.voxel_down_sample(pts, voxel_size)
        finally:
            tmp.close()

    def segment_floors_manually(self, path=None):
        """graph.py:624-787.  With a resident scene the whole step runs behind the C ABI (hmsg_segment_floors: the
        re-sampling and the height histogram on the device, scipy's filter / peak rules restated in C++), the map never
        leaves HBM and a storey's cloud is fetched only when somebody reads it; `_segment_floors_host` is the numpy /
        scipy mirror used for clouds loaded from disk (tests/test_floors_cabi.py: the two agree bit for bit)."""
        # a room level started early (start_room_level / create_feature_map) made floors already: a caller that drives this
        # method itself starts over -- the pending host stage is waited for and dropped with them
        early = getattr(self, "_room_level", None)
        if early is not None and not getattr(self, "_in_start_room_level", False):
            early["thread"].join()
            self._room_level = None
        if not getattr(self, "_in_start_room_level", False):
            for fl in self.floors:
                for r in fl.rooms:
                    if r in self.rooms:
                        self.rooms.remove(r)
            self.floors = []
        if self.scene is None or not isinstance(self.full_pcd, _LazyFn):     # (a cloud somebody set or loaded from disk)
            return self._segment_floors_host(path)
        floors = []
        for i, f in enumerate(self.scene.segment_floors()):
            lo, hi = float(f["y_lo"]), float(f["y_hi"])
            fl = Floor(str(i), name="floor_" + str(i))

            def crop(lo=lo, hi=hi):
                pts = self.full_pcd.points
                return pts[(pts[:, 1] >= lo) & (pts[:, 1] <= hi)]
            fl.pcd = _LazyFn(crop, int(f["n_points"]))
            fl._crop = (lo, hi)
            if f["n_points"]:
                mn, mx = np.asarray(f["bbox_min"]), np.asarray(f["bbox_max"])
                ex = mx - mn
                fl.vertices = np.array([mn, mn + [ex[0], 0, 0], mn + [0, ex[1], 0], mn + [0, 0, ex[2]], mx,
                                        mn + [0, ex[1], ex[2]], mn + [ex[0], 0, ex[2]], mn + [ex[0], ex[1], 0]])
            fl.floor_zero_level = float(f["zero_level"])
            fl.floor_height = float(f["height"])
            self.floors.append(fl)
            floors.append([lo, hi])
        return floors

    def _segment_floors_host(self, path=None):
        pts = self.full_pcd.points
        # graph.py:633: the map is re-sampled at 5 cm first.  It already holds one centroid per 5 cm voxel, but of a
        # grid with another origin (the new one hangs off the min bound of the FILTERED cloud), so neighbouring
        # centroids can fall into one new voxel and get averaged -- the histogram is taken over that cloud.
        down = self._voxel_down_sample(pts, 0.05)
        y = down[:, 1]
        bins = int(np.abs(np.max(y) - np.min(y)) / 0.01)
        hist = np.histogram(y, bins=bins)
        smooth = gaussian_filter1d(hist[0], sigma=2)                       # int64 in, int64 out (hazard 17)
        peaks, _ = find_peaks(smooth, distance=0.2 / 0.01, height=np.percentile(smooth, 90))
        locs = hist[1][peaks]
        # DBSCAN(eps=1, min_samples=1) on 1-D peak heights = chains of gaps <= 1 (graph.py:680-683)
        order = np.argsort(locs)
        labels = np.zeros(len(locs), dtype=int)
        lab = 0
        for a, b in zip(order[:-1], order[1:]):
            if locs[b] - locs[a] > 1:
                lab += 1
            labels[b] = lab
        # sklearn numbers clusters by first appearance in input order
        remap, nxt = {}, 0
        for l in labels:
            if l not in remap:
                remap[l] = nxt
                nxt += 1
        labels = np.array([remap[l] for l in labels], dtype=int)
        n_lab = len(np.unique(labels))
        clustered = []
        for i in range(n_lab):
            pk = peaks[labels == i]
            take = 1 if (i == 0 or i == n_lab - 1) else 2
            top = pk[np.argsort(smooth[pk])[-take:]].tolist()
            clustered.extend(hist[1][t] for t in top)
        clustered = np.sort(clustered)
        adjusted = []
        for i in range(len(clustered) - 1):
            adjusted.append(clustered[i])
            if clustered[i + 1] - clustered[i] >= 2.5:
                adjusted.append(clustered[i + 1] - 0.2)
        if len(clustered):
            adjusted.append(clustered[-1])
        floors = [[adjusted[i], adjusted[i + 1]] for i in range(len(adjusted) - 1)]
        if not floors:
            floors.append([float(hist[1].min()), float(hist[1].max())])
        floors[0][0] = (floors[0][0] + np.min(y)) / 2
        floors[-1][1] = np.max(y)
        yf = pts[:, 1]                      # the crop is taken from the full cloud (graph.py:769-775)
        for i, (lo, hi) in enumerate(floors):
            fl = Floor(str(i), name="floor_" + str(i))
            sel = pts[(yf >= lo) & (yf <= hi)]
            fl.pcd = _Pcd(sel)
            fl._crop = (float(lo), float(hi))     # the slab of the map this cloud is (room clouds on the device: hmsg_room_clouds)
            if len(sel):
                mn, mx = sel.min(0), sel.max(0)
                ex = mx - mn     # Open3D AABB corner order
                fl.vertices = np.array([mn, mn + [ex[0], 0, 0], mn + [0, ex[1], 0], mn + [0, 0, ex[2]], mx,
                                        mn + [0, ex[1], ex[2]], mn + [ex[0], 0, ex[2]], mn + [ex[0], ex[1], 0]])
                fl.floor_zero_level = float(np.min(sel[:, 1]))
            else:
                fl.floor_zero_level = float(lo)
            fl.floor_height = float(hi - fl.floor_zero_level)
            self.floors.append(fl)
        return floors

    # ------------------------------------------------------------------ A9 input
    def set_rooms(self, rooms: Sequence[dict]):
        """Rooms are an input (OpenCV watershed of graph.py:920-1189 is out of scope):
        dict(floor=int, vertices=[[x,z]...], name=str|None, view_frames=[frame idx...], view_embeddings=[[D]...])."""
        for spec in rooms:
            fl = self.floors[spec["floor"]]
            room = Room("%s_%d" % (fl.floor_id, len(fl.rooms)), fl.floor_id, name=spec.get("name"))
            room.vertices = np.asarray(spec["vertices"], dtype=np.float64)
            room.room_zero_level, room.room_height = fl.floor_zero_level, fl.floor_height
            room.embeddings = [np.asarray(e) for e in spec.get("view_embeddings", [])]
            for k, fidx in enumerate(spec.get("view_frames", [])):
                v = View("%s_%d" % (room.room_id, len(self.views)), room.room_id, img_id=int(fidx))
                room.views.append(v)
                self.views.append(v)
            fl.add_room(room)
            self.rooms.append(room)

    def set_view_feats(self, feats):
        """Global CLIP feature of every processed frame, [F, D] or list of [1, D] (graph.py:1119-1130 recomputes them
        with get_img_feats; they are the F_g the per-pixel stage already received, so loop B keeps them)."""
        self._view_feats = [np.asarray(f, np.float32).reshape(1, -1) for f in feats]

    @staticmethod
    def _room_extrusion(floor):
        """graph.py:1088-1103: the z levels of the extrusion (5 cm steps over the storey, negated) and T1 = the 90 degree turn
        about x as scipy's Rotation gives it (cos(90 deg) = 6.1e-17, not 0)."""
        from scipy.spatial.transform import Rotation
        z_levels = np.arange(floor.floor_zero_level, floor.floor_zero_level + floor.floor_height, 0.05).reshape(-1, 1)
        z_levels *= -1
        T1 = np.eye(4)
        T1[:3, :3] = Rotation.from_euler("x", 90, degrees=True).as_matrix()
        return z_levels, T1

    def _room_cloud(self, floor, floor_tree, room_m):
        """graph.py:1086-1108 on the host (a Graph without a resident scene): extrude the room's 2-D points over the floor's
        height in 5 cm steps, rotate into the map frame and pick the floor points nearest to them."""
        z_levels, T1 = self._room_extrusion(floor)
        room_m3d = np.concatenate([np.hstack((room_m, np.ones((room_m.shape[0], 1)) * z)) for z in z_levels], axis=0)
        X, Y, Z = room_m3d[:, 0], room_m3d[:, 1], room_m3d[:, 2]
        rows = [((X * T1[r, 0] + Y * T1[r, 1]) + Z * T1[r, 2]) + T1[r, 3] for r in range(4)]      # Open3D transform
        pts = np.stack([rows[0] / rows[3], rows[1] / rows[3], rows[2] / rows[3]], axis=1)
        _, idx = floor_tree.query(pts, k=1, workers=-1)
        mask = np.zeros(len(floor.pcd.points), bool)                      # Open3D select_by_index: unique, original order
        mask[idx] = True
        return _Pcd(np.asarray(floor.pcd.points)[mask])

    def _room_clouds_device(self, floor, room_2d_points):
        """The same for all rooms of a storey at once behind the C ABI (hmsg_room_clouds): nearest neighbours on the GPU,
        bit-equal ties by the restated cKDTree.  None when the floor cloud is not a slab of the resident map."""
        crop = getattr(floor, "_crop", None)
        if self.scene is None or crop is None:
            return None
        import time
        t0 = time.perf_counter()
        z_levels, T1 = self._room_extrusion(floor)
        sel, nf = self.scene.room_clouds(crop[0], crop[1], T1, z_levels, room_2d_points)
        t1 = time.perf_counter()
        n_floor = floor.pcd._n if isinstance(floor.pcd, _LazyFn) and floor.pcd._n is not None else len(np.asarray(floor.pcd.points))
        if nf != n_floor:
            return None
        # the clouds stay in HBM: a room's points are fetched (the storey's slab, then the selection) when somebody reads them
        out = [_LazyFn(lambda i=i: np.asarray(floor.pcd.points)[i], len(i)) for i in sel]
        self._room_clouds_resident = floor
        if os.environ.get("HMSG_DEBUG_TIMING"):
            print("[hmsg rooms] room clouds: library call %.1f ms, floor cloud + selection %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3),
                  file=sys.stderr)
        return out

    def _room_regions_device(self, floor):
        """graph.py:942-1084: the storey's rooms as lists of (x, z) cell centres -- histograms, blur / threshold /
        closings, distance transform, seeds and the watershed on the device (hmsg_segment_rooms; OpenCV restated, see
        oracle/rooms_oracle.py), then map_grid_to_point_cloud (graph_utils.py:359-388) per room.  None when the floor
        cloud is not a slab of the resident map."""
        crop = getattr(floor, "_crop", None)
        # (only when the floor cloud IS that slab of the resident map: a Graph with a resident scene whose full_pcd was set or
        #  loaded from disk gets `_crop` from _segment_floors_host too, and must segment the cloud it actually holds)
        if self.scene is None or crop is None or not isinstance(self.full_pcd, _LazyFn) or not isinstance(floor.pcd, _LazyFn):
            return None
        res = float(_get(self.cfg, "pipeline.grid_resolution", 0.05))
        markers, n_rooms, xz_min = self.scene.segment_rooms(crop[0], crop[1], floor.floor_zero_level, floor.floor_height, res)
        self.room_markers = getattr(self, "room_markers", {})
        self.room_markers[floor.floor_id] = markers
        out = []
        for i in range(n_rooms):
            y_cells, x_cells = np.where(markers == i + 1)
            out.append(np.column_stack(((x_cells - 10.5) * res + xz_min[0], (y_cells - 10.5) * res + xz_min[1])))
        return out

    def segment_hmsg_room(self, floor, path=None, room_2d_points=None, room_pcds=None):
        """graph.py:920-1189.  The rooms' 2-D regions (:942-1084, graph_utils.py:391-487: numpy + OpenCV in the
        reference) come from the device restatement (_room_regions_device; parity statistical) unless
        `room_2d_points` -- one [n, 2] array of (x, z) cell centres per room, what map_grid_to_point_cloud returns
        (:1084) -- is handed in.  Everything after it is mirrored exactly: room clouds (:1086-1108), camera -> room assignment and KMeans(24) representative views
        (compute_room_embeddings), Room nodes (:1147-1168) and one View node per (room, image) with the reference's id
        scheme -- `<floor>_<room>_<k>` with k counting across the floor's rooms, View.room_id the per-floor room
        INDEX (an int, :1176-1189)."""
        if isinstance(floor, (int, np.integer)):
            floor = self.floors[int(floor)]
        ctx = self._rooms_prepare(floor, room_2d_points, room_pcds)
        if ctx is None:
            return None
        self._rooms_embed(ctx, path)
        return self._rooms_finish(floor, ctx)

    # segment_hmsg_room in three stages, so that the middle one -- pure host work, scikit-learn's KMeans: ~35 ms per room -- can
    # run on a host thread beside the fusion and the merge fold (start_room_level); the result is the same either way.
    def _rooms_prepare(self, floor, room_2d_points=None, room_pcds=None):
        """Stage 1, device: regions (:942-1084), room clouds (:1086-1108), poses / global features of the processed frames
        (:1119-1136) and the camera -> room distance table (graph_utils.py:244-265)."""
        import time
        dbg = bool(os.environ.get("HMSG_DEBUG_TIMING"))
        t0 = time.perf_counter()
        if room_2d_points is None:
            room_2d_points = self._room_regions_device(floor)
        if room_2d_points is None:
            print("no room regions: the floor cloud is not a slab of the resident map (pass room_2d_points)")
            return None
        t1 = time.perf_counter()
        skip = int(_get(self.cfg, "pipeline.skip_frames", 1))
        if room_pcds is None:
            room_pcds = self._room_clouds_device(floor, [np.asarray(r, np.float64).reshape(-1, 2) for r in room_2d_points])
        t2 = time.perf_counter()
        if room_pcds is None:
            floor_pts = np.asarray(floor.pcd.points)
            tree = cKDTree(floor_pts)
            room_pcds = [self._room_cloud(floor, tree, np.asarray(r, np.float64).reshape(-1, 2)) for r in room_2d_points]
        ids = list(range(0, len(self.dataset), skip)) if self.dataset is not None else list(range(len(self._poses)))
        have = getattr(self, "_poses", None)
        if have is not None and len(have) == len(ids):                     # loop A kept them: no second read of every frame
            pose_list = [np.asarray(q, np.float64) for q in have]
        else:
            pose_list = [np.asarray(self.dataset[i][2], np.float64) for i in ids] if self.dataset is not None else list(self._poses)
        F_g_list = self._view_feats
        if len(F_g_list) != len(pose_list) and self.encoders is not None and self.dataset is not None:
            F_g_list = [np.asarray(self.encoders.extract(np.asarray(self.dataset[i][0]))["f_g"], np.float32).reshape(1, -1)
                        for i in ids]
        assert len(F_g_list) == len(pose_list), "one global feature per processed frame (set_view_feats)"
        if isinstance(floor.pcd, _LazyFn) and floor.pcd._n and np.asarray(floor.vertices).shape == (8, 3):
            # a slab of the resident map: its AABB came back with hmsg_segment_floors (vertices: min first, max fifth)
            pcd_min, pcd_max = np.asarray(floor.vertices[0], np.float64), np.asarray(floor.vertices[4], np.float64)
        else:
            floor_pts = np.asarray(floor.pcd.points)
            pcd_min, pcd_max = floor_pts.min(axis=0), floor_pts.max(axis=0)
        t3 = time.perf_counter()
        if getattr(self, "_room_clouds_resident", None) is floor and room_pcds and all(isinstance(r, _LazyFn) for r in room_pcds):
            # the clouds hmsg_room_clouds just made are still on the device: distances there, nothing travels
            cam = np.array([[pose[0, 3], pose[2, 3]] for pose in pose_list], dtype=np.float64).reshape(-1, 2)
            dist = self.scene.room_camera_distances(cam) if len(cam) else np.zeros((0, len(room_pcds)))
        else:
            dist = camera_room_distances(room_pcds, pose_list, lib=self.L, device_id=int(_get(self.cfg, "main.device_id", 0)))
        self._room_clouds_resident = None
        if dbg:
            print("[hmsg rooms] regions %.1f  room clouds %.1f  poses/feats %.1f  camera->room table %.1f ms" %
                  ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.perf_counter() - t3) * 1e3), file=sys.stderr)
        return dict(room_2d_points=room_2d_points, room_pcds=room_pcds, pose_list=pose_list, F_g_list=F_g_list, pcd_min=pcd_min,
                    pcd_max=pcd_max, dist=dist, skip=skip)

    @staticmethod
    def _rooms_embed(ctx, path=None):
        """Stage 2, host only (no handle, no Graph state): camera -> room assignment + KMeans(24) representative views."""
        ctx["emb"] = compute_room_embeddings(ctx["room_pcds"], ctx["pose_list"], ctx["F_g_list"], ctx["pcd_min"], ctx["pcd_max"], 24,
                                             path, dist=ctx["dist"])

    def _rooms_finish(self, floor, ctx):
        """Stage 3: Room nodes (:1147-1168) and one View node per (room, image) (:1176-1189)."""
        room_2d_points, room_pcds, skip = ctx["room_2d_points"], ctx["room_pcds"], ctx["skip"]
        repr_embs, repr_ids, room_id2img_id, room_clip = ctx["emb"]
        assert len(repr_embs) == len(room_2d_points) and len(room_id2img_id) == len(room_2d_points)
        self.room_id2img_ids = room_id2img_id
        for i in range(len(room_2d_points)):
            room = Room(str(floor.floor_id) + "_" + str(i), floor.floor_id, name="room_" + str(i))
            room.pcd = room_pcds[i] if isinstance(room_pcds[i], (_Pcd, _LazyFn)) or hasattr(type(room_pcds[i]), "points") else _Pcd(room_pcds[i])
            room.vertices = np.asarray(room_2d_points[i], np.float64).reshape(-1, 2)
            floor.add_room(room)
            room.room_height, room.room_zero_level = floor.floor_height, floor.floor_zero_level
            room.embeddings = repr_embs[i]
            room.represent_images = [int(k * skip) for k in repr_ids[i]]
            room.sample_images = [int(k * skip) for k in room_id2img_id[i]]
            room.clip_embeddings = room_clip[i]
            self.rooms.append(room)
        view_index = 0
        paths = getattr(self.dataset, "frameId2imgPath", None)
        for room_id in range(len(room_id2img_id)):
            for img_id in room_id2img_id[room_id]:
                retarget = img_id * skip
                view = View(str(floor.floor_id) + "_" + str(room_id) + "_" + str(view_index), room_id, retarget)
                view.img_path = paths[retarget] if paths is not None else None
                self.views.append(view)
                view_index += 1
                floor.rooms[room_id].views.append(view)
        return room_pcds

    def start_room_level(self):
        """The room level beside the fusion and the merge fold.  Floors (A8), room regions, room clouds and the camera -> room
        distance table (A9) need only the finished MAP and the frames' global features, not the instances: call this right
        after hmsg_finalize_map (create_feature_map does) -- the device work (a few ms) runs here, on the handle's stream, and
        the host-only rest (scikit-learn's KMeans, ~35 ms per room) on a host thread while the calling thread drives the
        fusion, the fold and the pooling (ctypes calls release the GIL).  build_hier_multimodal_scene_graph picks the result
        up; without this call it does the same work in place, with the same result."""
        import threading
        if self.floors or getattr(self, "_room_level", None) is not None:
            return
        import time
        t0 = time.perf_counter()
        self._in_start_room_level = True
        try:
            self.segment_floors_manually(None)
        finally:
            self._in_start_room_level = False
        t1 = time.perf_counter()
        ctxs = [self._rooms_prepare(fl) for fl in self.floors]
        if os.environ.get("HMSG_DEBUG_TIMING"):
            print("[hmsg rooms] floors %.1f ms, room level prepared in %.1f ms" % ((t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3), file=sys.stderr)
        box = dict(ctxs=ctxs, err=None)

        def work():
            try:
                for c in ctxs:
                    if c is not None:
                        self._rooms_embed(c)
            except BaseException as e:                                     # re-raised by the thread that joins
                box["err"] = e
        box["thread"] = threading.Thread(target=work, name="hmsg-room-level", daemon=True)
        box["thread"].start()
        self._room_level = box

    def set_label_feats(self, text_feats, classes):
        self._label_feats = (np.asarray(text_feats, np.float32), list(classes))

    def load_label_feats(self, obj_labels=None, label_dir=None):
        """The vocabulary segment_hmsg_objects names objects with (graph.py:1592-1597 get_label_feats): by name
        (cfg.pipeline.obj_labels; CSV sets are read from `label_dir` = cfg.pipeline.label_dir, the reference's
        memory/hmsg/labels) or a list of class names; features come from the 2-template text encoder or the cached
        text_feats_*.npy next to the CSV."""
        from .label_feats import get_label_feats
        obj_labels = obj_labels if obj_labels is not None else _get(self.cfg, "pipeline.obj_labels")
        label_dir = label_dir if label_dir is not None else _get(self.cfg, "pipeline.label_dir")
        text_feats, classes = get_label_feats(self.get_text_feats_multiple_templates, obj_labels, label_dir)
        self.set_label_feats(text_feats, classes)
        return text_feats, classes

    # ------------------------------------------------------------------ A10: graph.py:1582-1736
    def segment_hmsg_objects(self, save_dir=None):
        """graph.py:1582-1736.  With a resident scene the object -> floor / room / label assignment runs behind the C
        ABI (hmsg_build_object_nodes: per-object DBSCAN, find_intersection_share, label GEMM on the device); the
        per-view visibility test (check_object_in_view) needs the dataset's images and stays here."""
        want = _get(self.cfg, "pipeline.obj_labels")
        if self._label_feats is None and self.encoders is not None and want is not None and (
                not isinstance(want, str) or _get(self.cfg, "pipeline.label_dir") is not None):
            self.load_label_feats()
        text_feats, classes = self._label_feats if self._label_feats is not None else (None, None)
        if self.scene is not None:
            nodes = self.scene.build_object_nodes([f.floor_zero_level for f in self.floors], [f.floor_height for f in self.floors],
                                                  [int(r.floor_id) for r in self.rooms], [r.vertices for r in self.rooms], text_feats)
            store = _InstanceStore(self.scene)                               # points stay in HBM until somebody reads them
            self.mask_pcds = [_LazyPcd(store, i) for i in range(len(store.sizes))]
            picks = [(int(n["instance"]), self.rooms[int(n["room"])], int(n["label"])) for n in nodes]
        else:
            picks = self._assign_objects_host(text_feats)
        made = []
        for i, room, label in picks:
            pcd = self.mask_pcds[i]
            obj = Object(room.room_id + "_" + str(room.object_counter), room.room_id)
            room.object_counter += 1
            obj.name = classes[label] if (classes is not None and label >= 0) else "object"
            obj.pcd, obj.embedding = pcd, np.asarray(self.mask_feats[i]).reshape(-1)
            obj._instance = i if isinstance(pcd, _LazyPcd) else None       # (save_hmsg_graph: bulk writer)
            obj.vertices = None                                            # = points[:, [0, 2]], materialised on save
            made.append((obj, room, i))
        # view <-> object topology (:1712-1734).  A view's image is opened once (the reference opens it once per object) for
        # its size and pose; the test itself is numpy here, or -- pipeline.views_on_device -- one batch on the device
        # (hmsg_object_views) that never brings an object's points to the host.
        K, cams = None, {}
        def camera(img_id):
            if img_id not in cams:
                img, _, pose, _, _ = self.dataset[img_id]
                a = np.asarray(img)
                cams[img_id] = (a.shape[1], a.shape[0], np.linalg.inv(pose))
            return cams[img_id]
        # (object k, view) pairs in the reference's order -- objects in list order, each with its room's views in order --
        # as index arrays: ~10^5 pairs at 800 objects x 125 views a room
        usable = self.dataset is not None
        room_views = {}
        for _, room, _ in made:
            if id(room) not in room_views:
                room_views[id(room)] = [v for v in room.views if usable and v.img_id is not None]
        all_views, vpos = [], {}
        for vs in room_views.values():
            for v in vs:
                if id(v) not in vpos:
                    vpos[id(v)] = len(all_views)
                    all_views.append(v)
        room_vidx = {r: np.array([vpos[id(v)] for v in vs], np.int64) for r, vs in room_views.items()}
        counts = np.array([len(room_vidx[id(room)]) for _, room, _ in made], np.int64)
        pair_k = np.repeat(np.arange(len(made), dtype=np.int64), counts)
        pair_v = np.concatenate([room_vidx[id(room)] for _, room, _ in made]) if len(made) and counts.sum() else np.zeros(0, np.int64)
        best = {}
        if len(pair_k):
            K = self._K if self._K is not None else np.asarray(self.dataset.get_camera_intrinsics())
            view_img = np.array([v.img_id for v in all_views], np.int64)
            if self.scene is not None and bool(_get(self.cfg, "pipeline.views_on_device", False)):
                ids = np.unique(view_img)                                   # sorted image ids = camera table rows
                cam = [camera(int(i)) for i in ids]
                col_of_view = np.searchsorted(ids, view_img)
                inst = np.array([i for _, _, i in made], np.int64)
                vis, md = self.scene.object_views(np.stack([c[2] for c in cam]), [[c[0], c[1]] for c in cam], K,
                                                  inst[pair_k], col_of_view[pair_v])
            else:
                res = [check_object_in_view(*camera(all_views[v].img_id)[:2], K, camera(all_views[v].img_id)[2], made[k][0].pcd.points)
                       for k, v in zip(pair_k.tolist(), pair_v.tolist())]
                vis = np.array([r[0] for r in res], bool)
                md = np.array([r[1] for r in res], np.float64)
            hit = np.nonzero(vis)[0]
            hk, hv, hmd = pair_k[hit], pair_v[hit], md[hit]
            view_ids = [v.view_id for v in all_views]
            obj_ids = [obj.object_id for obj, _, _ in made]
            obj_names = [obj.name for obj, _, _ in made]
            # per object: its visible views in pair order (the pairs are sorted by object already); best view = the first
            # strictly smallest mean depth (:1727-1731)
            cut = np.searchsorted(hk, np.arange(len(made) + 1))
            hv_l = hv.tolist()
            for k in range(len(made)):
                a0, a1 = int(cut[k]), int(cut[k + 1])
                if a1 > a0:
                    made[k][0].view_ids.extend(view_ids[j] for j in hv_l[a0:a1])
                    # the reference scans with `if mean_depth < best_depth` from inf (:1727-1731): a NaN or inf mean never wins,
                    # and best_view_id stays None when no visible view has a finite one
                    seg = hmd[a0:a1]
                    fin = np.isfinite(seg)
                    if fin.any():
                        best[k] = (None, view_ids[hv_l[a0 + int(np.argmin(np.where(fin, seg, np.inf)))]])
            # per view: the objects that see it, in object order (a stable sort by view keeps the pair order inside a view)
            order = np.argsort(hv, kind="stable")
            sv, sk = hv[order], hk[order].tolist()
            vcut = np.searchsorted(sv, np.arange(len(all_views) + 1))
            for j, v in enumerate(all_views):
                a0, a1 = int(vcut[j]), int(vcut[j + 1])
                if a1 > a0:
                    v.object_ids.extend(obj_ids[k] for k in sk[a0:a1])
                    v.text_discription.extend(obj_names[k] for k in sk[a0:a1])
        for k, (obj, room, i) in enumerate(made):
            obj.best_view_id = best.get(k, (None, None))[1]
            room.add_object(obj)
            self.objects.append(obj)
        self._index = None

    def _assign_objects_host(self, text_feats):
        """The same assignment for a Graph without a resident scene (clouds loaded from disk): numpy / scipy."""
        names = None
        if text_feats is not None and len(self.mask_feats):
            emb = np.stack([np.asarray(f).reshape(-1) for f in self.mask_feats]).astype(np.float32)
            names = np.argmax(np.dot(emb.astype(np.float64), np.asarray(text_feats, np.float64).T), axis=1)
        margin, picks = 0.2, []
        room_centre = {r.room_id: np.mean(r.vertices, axis=0) for r in self.rooms}
        for fl in self.floors:
            for i, pcd in enumerate(self.mask_pcds):
                pts = np.asarray(pcd.points)
                if len(pts) < 10:
                    continue
                if not (pts[:, 1].min() > fl.floor_zero_level - margin and pts[:, 1].max() < fl.floor_zero_level + fl.floor_height + margin):
                    continue
                if not fl.rooms:
                    continue
                assoc = [find_intersection_share(r.vertices, pts[:, [0, 2]], 0.2) for r in fl.rooms]
                if np.sum(assoc) == 0:
                    c = np.mean(pts[:, [0, 2]], axis=0)
                    assoc = [-np.linalg.norm(room_centre[r.room_id] - c) for r in fl.rooms]
                picks.append((i, fl.rooms[int(np.argmax(assoc))], int(names[i]) if names is not None else -1))
        return picks

    def build_hier_multimodal_scene_graph(self, save_path=None, rooms: Sequence[dict] | None = None, room_regions=None):
        """graph.py:2033-2076 (navigation graph omitted).  Rooms: by default segment_hmsg_room per floor with the
        device room segmentation; `room_regions` = per floor a list of [n, 2] (x, z) region point arrays -> the same
        from the regions on (room clouds, room embeddings, View nodes); or `rooms` = ready-made room specs (set_rooms)."""
        early = getattr(self, "_room_level", None)
        if early is not None and rooms is None and room_regions is None:
            early["thread"].join()
            self._room_level = None
            if early["err"] is not None:
                raise early["err"]
            for fl, ctx in zip(self.floors, early["ctxs"]):
                if ctx is not None:
                    self._rooms_finish(fl, ctx)
            return self._build_hier_tail(save_path)
        if early is not None:                                              # (rooms handed in after all: drop the prepared ones)
            early["thread"].join()
            self._room_level = None
            self.floors = []
        self.segment_floors_manually(save_path)
        if room_regions is not None:
            for fl, regions in zip(self.floors, room_regions):
                self.segment_hmsg_room(fl, save_path, room_2d_points=regions)
        elif rooms is None and self.scene is not None:
            for fl in self.floors:                                         # graph.py:2044-2046
                self.segment_hmsg_room(fl, save_path)
        if rooms is not None:
            self.set_rooms(rooms)
        return self._build_hier_tail(save_path)

    def _build_hier_tail(self, save_path):
        self.segment_hmsg_objects(save_path)
        if _get(self.cfg, "pipeline.merge_objects_graph", False):          # graph.py:2053-2058 (false in every shipped config)
            for room in self.rooms:
                room.merge_objects(scene=self.scene)           # (the same-name overlap tests on the device when there is one)
            self.objects = [o for room in self.rooms for o in room.objects]
            self._index = None
        self.create_graph_new()
        if save_path is not None:
            self.save_hmsg_graph(os.path.join(save_path, "graph"))

    # ------------------------------------------------------------------ A11: graph.py:1752-1775
    def create_graph_new(self):
        """Building(0) - Floor - Room - Object, Room - View, View - Object.  As in the reference, `room.room_id ==
        view.room_id` compares the room's string id with the int room index a freshly built View carries
        (graph.py:1176-1189), so Room - View edges appear only after load_hmsg_graph."""
        for floor in self.floors:
            self.graph.add_node(floor, name="floor", type="floor")
            self.graph.add_edge(0, floor)
            for room in floor.rooms:
                self.graph.add_node(room, name="room", type="room")
                self.graph.add_edge(floor, room)
                for obj in room.objects:
                    self.graph.add_node(obj, name=obj.name, type="object")
                    self.graph.add_edge(room, obj)
        # (the reference walks ALL objects per view and tests `object_id in view.object_ids`: views x objects x list length
        #  string compares, seconds at 1000 views x 800 objects.  Same edges, same order -- per view its Room - View edges, then
        #  its objects in list order, each once -- from a position table, and handed to networkx in ONE call: the View nodes
        #  first (an edge adds no node here, so the node order is the reference's), then every edge in its order.)
        obj_pos = {}
        for k, obj in enumerate(self.objects):
            obj_pos.setdefault(obj.object_id, []).append(k)
        def view_edges(views):
            for view in views:
                for floor in self.floors:
                    for room in floor.rooms:
                        if room.room_id == view.room_id:
                            yield (room, view)
                            break                # (the reference's `break` leaves only the inner loop)
                for k in sorted({k for oid in view.object_ids for k in obj_pos.get(oid, ())}):
                    yield (view, self.objects[k])
        if all(self.graph.has_node(obj) for obj in self.objects):
            for view in self.views:
                self.graph.add_node(view, name="view", type="view")
            self.graph.add_edges_from(view_edges(self.views))
        else:                                    # an object outside every room would become a node by its first edge: view by view
            for view in self.views:
                self.graph.add_node(view, name="view", type="view")
                self.graph.add_edges_from(view_edges([view]))

    # ------------------------------------------------------------------ A11 persistence: graph.py:1801-1987
    def save_hmsg_graph(self, path):
        """graph.py:1801-1824: every node OF THE GRAPH is written (create_graph_new must have run)."""
        for sub in ("floors", "rooms", "objects", "views"):
            os.makedirs(os.path.join(path, sub), exist_ok=True)
        bulk = []
        for node in self.graph.nodes():
            # objects whose cloud and embedding still live in HBM (built by segment_hmsg_objects, untouched since) go
            # through the library's multi-threaded writer; everything else is written node by node
            if isinstance(node, Object) and self.scene is not None and getattr(node, "_instance", None) is not None \
                    and isinstance(node.pcd, _LazyPcd) and node.vertices is None:
                bulk.append(dict(instance=node._instance, object_id=node.object_id, room_id=node.room_id, name=node.name,
                                 view_ids=node.view_ids, best_view_id=node.best_view_id))
                continue
            for cls, sub in ((Floor, "floors"), (Room, "rooms"), (Object, "objects"), (View, "views")):
                if isinstance(node, cls):
                    if cls is Object:
                        node.save(os.path.join(path, sub))
                    else:                       # floors / rooms / views through the C ABI (hmsg_write_json / hmsg_write_ply)
                        node.save(os.path.join(path, sub), lib=self.L)
        if bulk:
            self.scene.save_objects(os.path.join(path, "objects"), bulk)

    save_graph = save_hmsg_graph

    def save_full_pcd(self, path):
        os.makedirs(path, exist_ok=True)
        _write_ply(os.path.join(path, "full_pcd.ply"), self.full_pcd.points)
        print("full pcd saved to disk in {}".format(path))

    def load_full_pcd(self, path):
        """graph.py:3782-3795."""
        if not os.path.exists(path):
            print("full pcd not found in {}".format(path))
            return None
        self.full_pcd = _Pcd(_read_ply(os.path.join(path, "full_pcd.ply")))
        print("full pcd loaded from disk with shape {}".format(np.asarray(self.full_pcd.points).shape))
        return self.full_pcd

    def save_full_pcd_feats(self, path):
        """graph.py:3797-3830: instances without points are dropped, then mask_feats.pt / full_feats.pt (torch)."""
        import torch
        os.makedirs(path, exist_ok=True)
        keep = [(p, f) for p, f in zip(self.mask_pcds, self.mask_feats) if len(p.points) > 0]
        self.mask_pcds, self.mask_feats = [k[0] for k in keep], [k[1] for k in keep]
        if len(self.mask_feats) != 0:
            self.mask_feats = np.array(self.mask_feats)
            torch.save(torch.from_numpy(self.mask_feats), os.path.join(path, "mask_feats.pt"))
        if len(self.full_feats_array) != 0:
            torch.save(torch.from_numpy(np.asarray(self.full_feats_array)), os.path.join(path, "full_feats.pt"))
        print("full pcd feats saved to disk in {}".format(path))

    def load_full_pcd_feats(self, path, full_feats=False, normalize=True):
        """graph.py:3832-3876."""
        import torch
        if not os.path.exists(path):
            print("full pcd feats not found in {}".format(path))
            return None
        t = torch.load(os.path.join(path, "full_feats.pt" if full_feats else "mask_feats.pt")).float()
        if normalize:
            t = torch.nn.functional.normalize(t, p=2, dim=-1)
        arr = t.cpu().numpy()
        if full_feats:
            self.full_feats_array = arr
        else:
            self.mask_feats = arr
        print("full pcd feats loaded from disk with shape {}".format(arr.shape))
        return arr

    def save_masked_pcds(self, path, state="both"):
        """graph.py:3880-3942: instances with fewer than 10 points go first; objects/pcd_<i>.ply (+ masked_pcd.ply)."""
        for i in reversed(range(len(self.mask_pcds))):
            if len(self.mask_pcds[i].points) < 10:
                self.mask_pcds.pop(i)
                self.mask_feats = list(self.mask_feats)
                self.mask_feats.pop(i)
        os.makedirs(path, exist_ok=True)
        objects_path = os.path.join(path, "objects")
        if state in ("both", "objects"):
            os.makedirs(objects_path, exist_ok=True)
            for i, pcd in enumerate(self.mask_pcds):
                _write_ply(os.path.join(objects_path, "pcd_%d.ply" % i), pcd.points)
        if state in ("both", "full"):
            allp = [np.asarray(p.points).reshape(-1, 3) for p in self.mask_pcds]
            _write_ply(os.path.join(path, "masked_pcd.ply"), np.concatenate(allp) if allp else np.zeros((0, 3)))
        print("masked pcds saved to disk in {}".format(path))

    def load_masked_pcds_new(self, path):
        """graph.py:3944-3990: needs the mask features first; features of missing clouds are deleted."""
        if len(self.mask_feats) == 0:
            print("load full pcd feats first")
            return None
        objects_path = os.path.join(path, "objects")
        if not os.path.exists(objects_path):
            print("masked pcds for objects not found in {}".format(path))
            return None
        self.mask_pcds, not_found = [], []
        for i in range(len(os.listdir(objects_path))):
            f = os.path.join(objects_path, "pcd_{}.ply".format(i))
            if os.path.exists(f):
                self.mask_pcds.append(_Pcd(_read_ply(f)))
            else:
                print("masked pcd {} not found in {}".format(i, path))
                not_found.append(i)
        not_found = [i for i in not_found if i < len(self.mask_feats)]
        self.mask_feats = np.delete(self.mask_feats, not_found, axis=0)
        return self.mask_pcds

    def load_hmsg_graph(self, path):
        if not os.path.isdir(path):
            print("graph not found in {}".format(path))
            return None
        self.graph_path = path
        self.floors, self.rooms, self.objects, self.views = [], [], [], []
        for f in sorted(x for x in os.listdir(os.path.join(path, "floors")) if x.endswith(".ply")):
            fl = Floor(f.split(".")[0], name="floor_" + f.split(".")[0])
            fl.load(os.path.join(path, "floors"))
            self.floors.append(fl)
            self.graph.add_node(fl, name="floor_" + f.split(".")[0], type="floor")
            self.graph.add_edge(0, fl)
        for f in sorted(x for x in os.listdir(os.path.join(path, "rooms")) if x.endswith(".ply")):   # lexicographic (:1931)
            rid = f.split(".")[0]
            room = Room(rid, rid.split("_")[0])
            room.load_new(os.path.join(path, "rooms"))
            self.rooms.append(room)
            self.graph.add_node(room, name="room_" + rid, type="room")
            self.graph.add_edge(self.floors[int(rid.split("_")[0])], room)
            fl = self.floors[int(room.floor_id)]
            if fl.rooms and isinstance(fl.rooms[0], str):
                fl.rooms = []
            fl.rooms.append(room)
        for f in sorted(x for x in os.listdir(os.path.join(path, "objects")) if x.endswith(".ply")):
            oid = f.split(".")[0]
            room_id = "_".join(oid.split("_")[:2])
            parent = next((r for r in self.rooms if r.room_id == room_id), None)
            assert parent is not None, f"Couldn't find the room with room id {room_id}"
            obj = Object(oid, room_id, name="object_" + oid)
            obj.load_new(os.path.join(path, "objects"))
            obj.room_id = room_id
            self.objects.append(obj)
            self.graph.add_node(obj, name="object_" + oid, type="object")
            self.graph.add_edge(parent, obj)
            parent.add_object(obj)
        for f in sorted(os.listdir(os.path.join(path, "views"))):
            vid = f.split(".")[0]
            v = View(vid, "_".join(vid.split("_")[:2]), name="view_" + vid)
            v.load(os.path.join(path, "views"))
            v.room_id = "_".join(vid.split("_")[:2])
            parent = next((r for r in self.rooms if r.room_id == v.room_id), None)
            assert parent is not None, f"Couldn't find the room with room id {v.room_id}"
            self.views.append(v)
            self.graph.add_node(v, name="view_" + vid, type="view")
            self.graph.add_edge(parent, v)
        self._index = None
        return self

    load_graph = load_hmsg_graph

    # ------------------------------------------------------------------ room names: graph.py:2129-2187
    def generate_room_names(self, generate_method="view_embedding", default_room_types=None):
        types = list(default_room_types or [])
        tf = self.get_text_feats_multiple_templates(types)
        for r in self.rooms:
            r.infer_room_type_from_view_embedding(types, tf)

    def set_room_names(self, room_names):
        for r, n in zip(self.rooms, room_names):
            r.name = n

    # ------------------------------------------------------------------ A12 queries
    def _node_index(self):
        if self._index is None:
            emb = np.stack([np.asarray(o.embedding, dtype=np.float64).reshape(-1) for o in self.objects])
            rid = {r.room_id: i for i, r in enumerate(self.rooms)}
            self._index = NodeIndex(emb, np.array([rid[o.room_id] for o in self.objects], np.int32), lib_=self.L)
        return self._index

    def query_floor(self, query, query_method="clip"):
        """graph.py:2216-2257."""
        order = np.argsort([f.floor_zero_level for f in self.floors])
        try:
            return order[int(query) - 1]
        except BaseException:
            t = self.get_text_feats_multiple_templates([query])
            names = self.get_text_feats_multiple_templates(["floor " + str(i) for i in range(len(self.floors))])
            return order[np.argsort(np.dot(t, names.T)[0])[::-1][0]]

    def query_hmsg_room(self, query, floor_id=-1, query_method="view_embedding"):
        """graph.py:3164-3272."""
        valid = query is not None and query != "" and "unknown" not in query.lower()
        t = self.get_text_feats_multiple_templates([query])
        rooms = self.rooms if floor_id == -1 else self.floors[floor_id].rooms
        if query_method == "label" and valid:
            for r in rooms:
                assert r.name is not None, "The name attribute for the room has not been generated"
            sim = np.dot(t, self.get_text_feats_multiple_templates([r.name for r in rooms]).T)
            order = np.argsort(sim[0])[::-1]
            keep = [order[0]] + [i for i in order[1:] if np.abs(sim[0, i] - sim[0, order[0]]) < 1e-3]
            ids = {rooms[i].room_id for i in keep}
            return [i for i, r in enumerate(rooms) if r.room_id in ids]
        sims = {}
        for r in rooms:
            sims[r.room_id] = np.max(np.dot(t, np.stack(r.embeddings).T))
        ranked = {int(k.split("_")[-1]): v for k, v in sorted(sims.items(), key=lambda kv: kv[1], reverse=True)}
        return list(ranked.keys())[: min(len(ranked), 5 if valid else 10)]

    def query_hmsg_object(self, query, floor_id=-1, room_ids=[], query_method="clip", top_k=1, negative_prompt=[]):
        """graph.py:3056-3162; the similarity GEMM, class arg-max and top-k run on the GPU."""
        if query in negative_prompt:
            qid, cats = negative_prompt.index(query), list(negative_prompt)
        else:
            qid, cats = 0, [query, *negative_prompt]
        t = self.get_text_feats_multiple_templates(cats)
        gl = {r.room_id: i for i, r in enumerate(self.rooms)}
        if floor_id != -1:
            rooms_global = [gl[self.floors[floor_id].rooms[i].room_id] for i in room_ids]
        else:
            rooms_global = list(room_ids)
        idx, room, score = self._node_index().query_objects(t[None], np.array([qid], np.int32), [rooms_global], top_k,
                                                            use_negatives=len(negative_prompt) > 0)
        keep = idx[0] >= 0
        back = {g: l for l, g in zip(room_ids, rooms_global)}
        return [int(i) for i in idx[0][keep]], [back[int(r)] for r in room[0][keep]], [float(s) for s in score[0][keep]]

    def rank_goal_views(self, object_query, rooms_list, top_k=24):
        """The view search of the slow path (graph.py:2864-2897): similarity of the object query to the CLIP embedding of
        EVERY sampled image of the candidate rooms, arg-max and the top-24 (np.argsort(sims)[-k:][::-1]).  The similarity
        matrix comes from the device (float64 MFMA GEMM); the ordering is numpy's.  Returns (best image id, top image
        ids, similarities)."""
        img_ids, embs = [], []
        for room in rooms_list:
            assert len(room.sample_images) == len(room.clip_embeddings), \
                f"Number of images ({len(room.sample_images)}) != embeddings ({len(room.clip_embeddings)})"
            img_ids.extend(room.sample_images)
            embs.extend(np.asarray(e).reshape(-1) for e in room.clip_embeddings)
        if not img_ids:
            return None, [], np.zeros(0)
        t = self.get_text_feats_multiple_templates([object_query])
        table = np.ascontiguousarray(np.stack(embs), dtype=np.float64)
        ix = NodeIndex(table, np.zeros(len(table), np.int32), lib_=self.L)
        sims = ix.similarity(t[:1])[0]
        ix.close()
        k = min(top_k, sims.shape[0])
        top_idx = np.argsort(sims)[-k:][::-1]
        return img_ids[int(np.argmax(sims))], [img_ids[int(i)] for i in top_idx], sims

    def _parse(self, query_instruction):
        if isinstance(query_instruction, str):
            return type(self).instruction_parser(query_instruction)
        return tuple(query_instruction)                     # already parsed (floor | None, room, object)

    def _hier_query(self, query_instruction, top_k, negatives, icra):
        import time
        t0 = time.time()
        floor_q, room_q, obj_q = self._parse(query_instruction)
        llm_parse_time = time.time() - t0
        if icra and room_q and "Exhibition" in room_q:
            negatives = ["wall"]
        floor_id = self.query_floor(floor_q) if floor_q is not None else -1
        if room_q is not None and obj_q is not None:
            # room stage + object stage on the device in one call (include/hmsg.h: hmsg_query_hier)
            (room_ids, obj_ids, room_ids2, scores), = self.query_hierarchy_batch([(floor_id, room_q, obj_q, negatives)], top_k)
        else:
            room_ids = self.query_hmsg_room(room_q, floor_id=floor_id, query_method="label") if room_q is not None else []
            obj_ids, room_ids2, scores = self.query_hmsg_object(obj_q, floor_id=floor_id, room_ids=room_ids, top_k=top_k,
                                                                negative_prompt=negatives) if obj_q is not None else ([], [], [])
        res = dict(room_query=room_q, object_query=obj_q, negative_labels=negatives, object_scores=scores)
        if icra:
            res.update(LLM_Parse_Time=llm_parse_time, FastMatching=0.0, ObjectInImageCheck=0.0, VLM_Rethinking=0.0,
                       Re_Matching=0.0, Total_Time=0.0)
        rooms = [self.floors[floor_id].rooms[k] for k in room_ids2] if floor_id != -1 else [self.rooms[k] for k in room_ids2]
        return (self.floors[floor_id] if floor_id != -1 else None, rooms, [self.objects[i] for i in obj_ids], res)

    def _hier_index(self):
        """the node index with the levels above it resident (room name / view embeddings, floors -> rooms)"""
        ix = self._node_index()
        # The upper levels live ON the index object: a rebuilt index (segment_hmsg_objects, merge_objects, load_hmsg_graph
        # reset self._index) has none, and new room names / rooms / view embeddings make the resident ones stale.  The
        # signature is what set_hierarchy consumes, so any change to it re-uploads.
        sig = (tuple(str(r.room_id) for r in self.rooms), tuple(getattr(r, "name", None) for r in self.rooms),
               tuple(len(getattr(r, "embeddings", []) or []) for r in self.rooms),
               tuple(id(getattr(r, "embeddings", None)) for r in self.rooms),
               tuple(tuple(str(r.room_id) for r in f.rooms) for f in self.floors))
        if getattr(ix, "_hier_sig", None) != sig:
            gl = {r.room_id: i for i, r in enumerate(self.rooms)}
            named = all(getattr(r, "name", None) is not None for r in self.rooms)
            names = self.get_text_feats_multiple_templates([r.name for r in self.rooms]) if named and self.rooms else None
            views = [np.stack(r.embeddings) if len(getattr(r, "embeddings", []) or []) else np.zeros((0, ix.D)) for r in self.rooms]
            ix.set_hierarchy([[gl[r.room_id] for r in f.rooms] for f in self.floors], names, views,
                             [int(str(r.room_id).split("_")[-1]) for r in self.rooms])
            ix._hier_sig = sig
        return ix

    def query_hierarchy_batch(self, queries, top_k=1):
        """queries: (floor_id, room_query, object_query, negative_prompt) tuples -> per query (room numbers as
        query_hmsg_room(..., "label") returns them, object indices, their room numbers, scores) -- graph.py:3538-3568 for a
        whole batch, floor -> room -> object on the device.  Queries are grouped by the number of text rows."""
        ix = self._hier_index()
        out = [None] * len(queries)
        groups = {}
        for n, (floor_id, room_q, obj_q, negs) in enumerate(queries):
            if obj_q in negs:
                qid, cats = negs.index(obj_q), list(negs)
            else:
                qid, cats = 0, [obj_q, *negs]
            groups.setdefault((len(cats), len(negs) > 0), []).append((n, floor_id, room_q, qid, cats))
        for (_, use_neg), items in groups.items():
            T = np.stack([self.get_text_feats_multiple_templates(cats) for _, _, _, _, cats in items])
            Tr = np.stack([self.get_text_feats_multiple_templates([room_q])[0] for _, _, room_q, _, _ in items])
            fl = np.array([f for _, f, _, _, _ in items], np.int32)
            valid = [rq is not None and rq != "" and "unknown" not in rq.lower() for _, _, rq, _, _ in items]
            mode = np.array([1 if v else 3 for v in valid], np.int32)       # an invalid room text falls into the view branch (top 10)
            sel, idx, room, score = ix.query_hier(T, np.array([q for _, _, _, q, _ in items], np.int32), Tr, fl, mode, top_k,
                                                  use_negatives=use_neg)
            for j, (n, floor_id, _, _, _) in enumerate(items):
                rooms_list = list(range(len(self.rooms))) if floor_id == -1 else \
                    [next(i for i, r in enumerate(self.rooms) if r.room_id == fr.room_id) for fr in self.floors[floor_id].rooms]
                back = {rooms_list[p]: p for p in sel[j]}
                keep = idx[j] >= 0
                out[n] = (sel[j], [int(i) for i in idx[j][keep]], [back[int(r)] for r in room[j][keep]], [float(v) for v in score[j][keep]])
        return out

    def query_hierarchy_protected_icra(self, query_instruction, top_k=1, use_gpt=False):
        """graph.py:3483-3591: negatives ["background"] (["wall"] for an "Exhibition" room); `query_instruction` is the
        sentence (parsed by Graph.instruction_parser, the reference asks an LLM) or an already parsed triple.  The
        slow-reasoning branch (use_gpt) is out of scope."""
        return self._hier_query(query_instruction, top_k, ["background"], icra=True)

    def query_hierarchy_protected(self, query_instruction, top_k=1, use_gpt=False):
        """graph.py:3593-3716: the long background-label list + monitor / wall / speaker as negatives."""
        return self._hier_query(query_instruction, top_k, BACKGROUND_LABELS + ["monitor", "wall", "speaker"], icra=False)

    # ------------------------------------------------------------------ assembly on an already built Scene
    @classmethod
    def from_scene(cls, scene: Scene, cfg=None, encoders=None, lib: HmsgLib | None = None, instances=True):
        """Wrap a Scene whose A1..A7 stages already ran (bench.py / services that drive the C ABI directly):
        the returned Graph can run build_hier_multimodal_scene_graph (A8, A10, A11) and the queries.
        instances=False: only the MAP is final so far (hmsg_finalize_map) -- enough for start_room_level; call
        take_instances() once the merge and the pooling have run."""
        g = cls(cfg or dict(main=dict(), models=dict(clip=dict(feat_dim=scene.cfg.feat_dim))), encoders=encoders,
                lib=lib or scene.L)
        g.scene = scene
        g.full_pcd = _LazyFn(scene.map_points, scene.map_size())     # stays in HBM until read
        if instances:
            g.take_instances()
        return g

    def take_instances(self):
        """the pooled instance features of the resident scene (after hmsg_pool_instances)"""
        self.mask_feats = list(self.scene.instance_feats())
        self.mask_pcds = [None] * len(self.mask_feats)
