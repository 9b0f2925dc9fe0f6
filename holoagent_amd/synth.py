"""Seeded synthetic posed-RGB-D scene generator (input data only, not part of the hot path).

Produces the tuple contract of the reference datasets
(`fsr_vln/memory/hmsg/dataloader/hm3dsem.py:45-155`): per frame an rgb u8 [H,W,3],
a depth u16 [H,W] in millimetres (scale 1000, `hm3dsem.py:40`), a camera-to-world
pose f64 [4,4] and the pin-hole intrinsics of `hm3dsem.py:140-155` (hfov 90 deg),
plus what the wrapped encoders would hand over for that frame: M boolean masks
(`mask_generator.generate(...)[i]["segmentation"]`) and the CLIP features
F_g [D], F_masked [M,D], F_crop [M,D] (unit f32 rows) consumed by
`perception/models/sam_clip_feats_extractor.py:159-175`.

Scene: axis-aligned box rooms on a grid, up = +y, K object
boxes per room; the camera stands inside one (closed) room per frame and yaws in
fixed steps; depth = z of the first ray/box hit.
"""
from __future__ import annotations

import dataclasses
import numpy as np


@dataclasses.dataclass
class SceneSpec:
    seed: int = 1234
    rooms_x: int = 4
    rooms_z: int = 2
    room_size: tuple = (5.0, 3.0, 4.0)        # (x, y=up, z)
    objects_per_room: int = 8
    width: int = 640
    height: int = 480
    n_frames: int = 100
    n_masks: int = 32
    feat_dim: int = 512
    yaw_step_deg: float = 10.0
    cam_height: float = 1.5
    depth_cut: float = 10.0
    feat_noise: float = 0.02
    depth_noise_mm: float = 1.5      # sensor noise (std, mm) added before the u16 quantisation


class SynthScene:
    """Deterministic scene; `frame(i)` renders one posed RGB-D frame with masks and features."""

    def __init__(self, spec: SceneSpec):
        self.spec = spec
        rng = np.random.Generator(np.random.PCG64(spec.seed))
        sx, sy, sz = spec.room_size
        rooms = []
        objects = []           # (room_id, lo[3], hi[3])
        for rz in range(spec.rooms_z):
            for rx in range(spec.rooms_x):
                lo = np.array([rx * sx, 0.0, rz * sz])
                hi = lo + np.array([sx, sy, sz])
                rid = len(rooms)
                rooms.append((lo, hi))
                for _ in range(spec.objects_per_room):
                    size = rng.uniform(0.3, 1.5, size=3)
                    size[1] = rng.uniform(0.3, 1.2)
                    # keep clear of the room centre column where the camera stands
                    for _try in range(64):
                        c = np.array([rng.uniform(lo[0] + 0.2 + size[0] / 2, hi[0] - 0.2 - size[0] / 2),
                                      0.0,
                                      rng.uniform(lo[2] + 0.2 + size[2] / 2, hi[2] - 0.2 - size[2] / 2)])
                        centre = (lo + hi) / 2
                        if abs(c[0] - centre[0]) > size[0] / 2 + 0.45 or abs(c[2] - centre[2]) > size[2] / 2 + 0.45:
                            break
                    base_y = 0.0 if rng.random() < 0.7 else rng.uniform(0.5, 1.5)
                    olo = np.array([c[0] - size[0] / 2, base_y, c[2] - size[2] / 2])
                    ohi = np.array([c[0] + size[0] / 2, min(base_y + size[1], sy - 0.1), c[2] + size[2] / 2])
                    objects.append((rid, olo, ohi))
        self.rooms = rooms
        self.objects = objects
        n_obj = len(objects)
        n_faces = len(rooms) * 6
        D = spec.feat_dim
        u = rng.standard_normal((n_obj + n_faces, D)).astype(np.float32)
        u /= np.linalg.norm(u, axis=1, keepdims=True)
        self.entity_feats = u                      # objects first, then room faces
        self.n_obj = n_obj
        hfov = np.pi / 2
        self.K = np.array([[spec.width / (2 * np.tan(hfov / 2)), 0, spec.width / 2],
                           [0, spec.width / (2 * np.tan(hfov / 2)), spec.height / 2],
                           [0, 0, 1.0]])
        self._rng_seed = spec.seed

    # ------------------------------------------------------------------ poses
    def pose(self, i: int) -> np.ndarray:
        """Camera-to-world 4x4 (camera: x right, y down, z forward; world: y up)."""
        spec = self.spec
        n_rooms = len(self.rooms)
        per_room = max(1, int(np.ceil(spec.n_frames / n_rooms)))
        rid = min(i // per_room, n_rooms - 1)
        k = i - rid * per_room
        lo, hi = self.rooms[rid]
        c = (lo + hi) / 2
        # small deterministic wander around the room centre
        pos = np.array([c[0] + 0.25 * np.sin(0.37 * k), spec.cam_height + 0.05 * np.sin(0.11 * k),
                        c[2] + 0.25 * np.cos(0.23 * k)])
        yaw = np.deg2rad(spec.yaw_step_deg * k + 3.0 * rid)
        pitch = np.deg2rad(8.0 * np.sin(0.31 * k))
        fwd = np.array([np.sin(yaw) * np.cos(pitch), -np.sin(pitch), np.cos(yaw) * np.cos(pitch)])
        up = np.array([0.0, 1.0, 0.0])
        right = np.cross(up, fwd)
        right /= np.linalg.norm(right)
        down = np.cross(fwd, right)
        T = np.eye(4)
        T[:3, 0] = right
        T[:3, 1] = down
        T[:3, 2] = fwd
        T[:3, 3] = pos
        return T, rid

    # ----------------------------------------------------------------- render
    def frame(self, i: int):
        spec = self.spec
        H, W = spec.height, spec.width
        T, rid = self.pose(i)
        R = T[:3, :3]
        o = T[:3, 3]
        K = self.K
        ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
        dc = np.stack([(xs + 0.5 - K[0, 2]) / K[0, 0], (ys + 0.5 - K[1, 2]) / K[1, 1],
                       np.ones_like(xs, dtype=np.float64)], axis=-1)      # camera rays with z=1
        dw = dc @ R.T                                                      # world dirs (z-depth parametrised)
        inv = 1.0 / np.where(np.abs(dw) < 1e-12, 1e-12, dw)
        lo, hi = self.rooms[rid]
        t1 = (lo - o) * inv
        t2 = (hi - o) * inv
        tfar_axis = np.maximum(t1, t2)
        t_room = tfar_axis.min(axis=-1)
        face_axis = tfar_axis.argmin(axis=-1)
        face_side = (np.take_along_axis(dw, face_axis[..., None], -1)[..., 0] > 0).astype(np.int64)
        ent = self.n_obj + rid * 6 + face_axis * 2 + face_side            # entity id per pixel
        depth = t_room.copy()
        for oid, (orid, olo, ohi) in enumerate(self.objects):
            if orid != rid:
                continue
            a = (olo - o) * inv
            b = (ohi - o) * inv
            tn = np.minimum(a, b).max(axis=-1)
            tf = np.maximum(a, b).min(axis=-1)
            hit = (tn < tf) & (tn > 0.05) & (tn < depth)
            depth = np.where(hit, tn, depth)
            ent = np.where(hit, oid, ent)
        rng_d = np.random.Generator(np.random.PCG64([self._rng_seed, 15485863, i]))
        depth_mm = np.rint(depth * 1000.0 + spec.depth_noise_mm * rng_d.standard_normal(depth.shape))
        depth_mm[(depth > spec.depth_cut) | (depth_mm > 65535) | (depth_mm < 1)] = 0
        depth_u16 = depth_mm.astype(np.uint16)
        # colours: hash of the entity id
        col = np.stack([(ent * 53 + 17) % 256, (ent * 97 + 101) % 256, (ent * 193 + 7) % 256], -1).astype(np.uint8)
        # masks: largest visible entities first, padded with rectangular tiles (overlaps allowed)
        M = spec.n_masks
        ids, counts = np.unique(ent, return_counts=True)
        order = np.argsort(-counts, kind="stable")
        ids = ids[order][: max(0, M - 4)]
        masks = np.zeros((M, H, W), dtype=bool)
        mask_ent = np.zeros(M, dtype=np.int64)
        m = 0
        for e in ids:
            masks[m] = ent == e
            mask_ent[m] = e
            m += 1
        rng = np.random.Generator(np.random.PCG64([self._rng_seed, 7919, i]))
        while m < M:
            h = int(rng.integers(H // 8, H // 3))
            w = int(rng.integers(W // 8, W // 3))
            y0 = int(rng.integers(0, H - h))
            x0 = int(rng.integers(0, W - w))
            masks[m, y0:y0 + h, x0:x0 + w] = True
            # a tile "sees" the dominant entity under it
            vals, cnts = np.unique(ent[y0:y0 + h, x0:x0 + w], return_counts=True)
            mask_ent[m] = vals[np.argmax(cnts)]
            m += 1
        D = spec.feat_dim
        u = self.entity_feats[mask_ent]
        f_masked = u + spec.feat_noise * rng.standard_normal((M, D)).astype(np.float32)
        f_crop = u + spec.feat_noise * rng.standard_normal((M, D)).astype(np.float32)
        f_masked /= np.linalg.norm(f_masked, axis=1, keepdims=True)
        f_crop /= np.linalg.norm(f_crop, axis=1, keepdims=True)
        f_g = u.mean(axis=0)
        f_g /= np.linalg.norm(f_g)
        return dict(rgb=col, depth=depth_u16, pose=T, K=K.copy(), masks=masks,
                    f_g=f_g.astype(np.float32)[None, :], f_masked=f_masked.astype(np.float32),
                    f_crop=f_crop.astype(np.float32), room=rid, mask_entity=mask_ent)

    # --------------------------------------------------------------- queries
    def text_table(self, n_queries: int, noise: float = 0.1):
        """Per query a (query, negative) pair of 'template-averaged' text features [n,2,D] (not unit norm,
        like `clip_utils.py:346-347`) and the entity each query describes."""
        rng = np.random.Generator(np.random.PCG64([self._rng_seed, 104729]))
        D = self.spec.feat_dim
        ents = rng.integers(0, self.n_obj, size=n_queries)
        out = np.zeros((n_queries, 2, D), dtype=np.float32)
        neg = rng.standard_normal(D).astype(np.float32)
        neg /= np.linalg.norm(neg)
        for q, e in enumerate(ents):
            t = []
            for _tmpl in range(2):
                v = self.entity_feats[e] + noise * rng.standard_normal(D).astype(np.float32)
                t.append(v / np.linalg.norm(v))
            out[q, 0] = np.mean(np.stack(t), axis=0)
            out[q, 1] = neg
        return out, ents
