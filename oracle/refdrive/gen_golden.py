"""Golden-vector generator: runs the REFERENCE's own Python (imported from /root/reference, which
exists only in the build container) on seeded synthetic inputs and stores inputs + outputs as small
.npz fixtures under tests/golden/.  Nothing of the reference's source travels: only data.

    python -m oracle.refdrive.gen_golden            # from the repo root

Import recipe: SURVEY.md section 8c (MagicMock the absent third-party modules, inject the numpy
Open3D / faiss stand-ins of `fake_backends.py`, pin torch to one thread so the duplicate-index `+=`
of graph.py:410 is last-writer-wins).
"""
from __future__ import annotations

import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
REF = "/root/reference/fsr_vln"


def import_reference():
    from oracle.refdrive import fake_backends as FB
    for m in ["cv2", "torchmetrics", "torchmetrics.functional", "open_clip", "segment_anything", "oss2",
              "oss2.credentials", "openai", "hydra", "omegaconf", "torchvision", "pyvista", "skfmm", "plyfile"]:
        sys.modules[m] = MagicMock()
    sys.modules["open3d"] = FB.make_open3d()
    sys.modules["faiss"] = FB.make_faiss()
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.switch_backend("Agg")                       # load a head-less backend, then pin it (navigation_graph.py:33 asks for TkAgg)
    plt.switch_backend = lambda *a, **k: None
    sys.path.insert(0, REF)
    import torch
    torch.set_num_threads(1)
    torch.Tensor.cuda = lambda self, *a, **k: self
    import memory.hmsg.graph.graph as G
    import perception.models.sam_clip_feats_extractor as X

    class TorchProxy:
        def __getattr__(self, name):
            return getattr(torch, name)

        @staticmethod
        def zeros(*a, device=None, **k):
            return torch.zeros(*a, **k)

    X.torch = TorchProxy()
    return G, X


class AttrDict(dict):
    __getattr__ = dict.__getitem__


def make_dataset(G, frames):
    from PIL import Image
    from memory.hmsg.dataloader.generic import RGBDDataset

    class SynthDataset(RGBDDataset):
        def __init__(self, frames):
            self.frames = frames
            self.depth_intrinsics = frames[0]["K"]
            self.scale = 1000.0
            self.data_list = list(range(len(frames)))

        def _get_data_list(self):
            return self.data_list

        def __getitem__(self, i):
            f = self.frames[i]
            return Image.fromarray(f["rgb"]), Image.fromarray(f["depth"]), f["pose"], None, f["K"]

        def get_camera_intrinsics(self):
            return self.depth_intrinsics

    return SynthDataset(frames)


def drive_create_feature_map(G, X, frames, cfg):
    """Run the reference's Graph.create_feature_map (graph.py:262-491) on synthetic frames."""
    state = dict(i=-1, phase=0)

    class MaskGen:
        def generate(self, image):
            state["i"] += 1
            fr = frames[state["i"]]
            return [dict(segmentation=fr["masks"][m], predicted_iou=1.0, bbox=[0, 0, 1, 1])
                    for m in range(fr["masks"].shape[0])]

    X.get_img_feats = lambda img, pre, model: frames[state["i"]]["f_g"].copy()
    X.crop_all_bounding_boxs = lambda image, masks, block_background, bbox_margin: \
        [("masked" if block_background else "crop", m) for m in range(len(masks))]

    def batch(imgs, pre, model):
        fr = frames[state["i"]]
        return (fr["f_masked"] if imgs[0][0] == "masked" else fr["f_crop"]).copy()

    X.get_img_feats_batch = batch
    X.cv2 = MagicMock()

    g = G.Graph.__new__(G.Graph)
    g.cfg = AttrDict(main=AttrDict(save_path="/tmp/hmsg_golden_tmp"),
                     pipeline=AttrDict(**cfg))
    g.full_pcd = sys.modules["open3d"].geometry.PointCloud()
    g.mask_feats, g.mask_pcds, g.full_feats_array = [], [], []
    g.dataset = make_dataset(G, frames)
    g.mask_generator = MaskGen()
    g.clip_model = None
    g.preprocess = None
    g.clip_feat_dim = cfg["feat_dim"]
    g.save_full_pcd = lambda path=None: None
    g.create_feature_map()
    return g


def pack_clouds(clouds):
    pts = [np.asarray(c.points).reshape(-1, 3) for c in clouds]
    off = np.cumsum([0] + [len(p) for p in pts])
    return (np.concatenate(pts) if pts else np.zeros((0, 3))), off


def pack_frames(frames):
    """Frames may hold different numbers of masks: rows are padded to the largest count, `n_masks` keeps the counts."""
    M = max(f["masks"].shape[0] for f in frames)

    def pad(a):
        out = np.zeros((M,) + a.shape[1:], a.dtype)
        out[: a.shape[0]] = a
        return out
    return dict(
        rgb=np.stack([f["rgb"] for f in frames]), depth=np.stack([f["depth"] for f in frames]),
        pose=np.stack([f["pose"] for f in frames]), K=frames[0]["K"],
        masks=np.packbits(np.stack([pad(f["masks"]) for f in frames]), axis=-1),
        f_g=np.stack([f["f_g"] for f in frames]), f_masked=np.stack([pad(f["f_masked"]) for f in frames]),
        f_crop=np.stack([pad(f["f_crop"]) for f in frames]),
        n_masks=np.array([f["masks"].shape[0] for f in frames], np.int32))


def ragged_frames(spec, counts):
    """Frames whose mask count differs from frame to frame, like SAM's output (1 .. more than 64)."""
    from holoagent_amd.synth import SynthScene
    sc = SynthScene(spec)
    frames = []
    for i in range(spec.n_frames):
        fr = sc.frame(i)
        m = int(counts[i % len(counts)])
        fr["masks"], fr["f_masked"], fr["f_crop"] = fr["masks"][:m], fr["f_masked"][:m], fr["f_crop"][:m]
        frames.append(fr)
    return frames


def gen_build(G, X, out_dir, only=None):
    from holoagent_amd.synth import SceneSpec, SynthScene
    cases = [
        ("build_seq", SceneSpec(seed=4321, rooms_x=1, rooms_z=1, room_size=(4.0, 2.6, 3.5), objects_per_room=5,
                                width=160, height=120, n_frames=36, n_masks=12, feat_dim=32), "sequential", None),
        ("build_hier", SceneSpec(seed=99, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4,
                                 width=128, height=96, n_frames=30, n_masks=10, feat_dim=16,
                                 yaw_step_deg=12.0), "hierarchical", None),
        # SAM returns a different number of masks per frame (here 1 .. 70, i.e. also more than one 64-bit word)
        ("build_ragged", SceneSpec(seed=777, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4,
                                   width=128, height=96, n_frames=24, n_masks=70, feat_dim=24, yaw_step_deg=14.0),
         "sequential", [70, 1, 33, 5, 64, 65, 2, 40, 12, 3, 66, 9]),
    ]
    for name, spec, merge_type, counts in cases:
        if only and name not in only:
            continue
        if counts is None:
            sc = SynthScene(spec)
            frames = [sc.frame(i) for i in range(spec.n_frames)]
        else:
            frames = ragged_frames(spec, counts)
        cfg = dict(voxel_size=0.05, skip_frames=1, init_overlap_thresh=0.75, overlap_thresh_factor=0.025,
                   iou_thresh=0.05, clip_masked_weight=0.4418, clip_bbox_margin=50, max_mask_distance=10000,
                   merge_type=merge_type, feat_dim=spec.feat_dim)
        g = drive_create_feature_map(G, X, frames, cfg)
        cloud = np.asarray(g.full_pcd.points)
        mp, moff = pack_clouds(g.mask_pcds)
        feats = np.stack([np.asarray(f).reshape(-1) for f in g.mask_feats]) if g.mask_feats else np.zeros((0, 1))
        # Which instances hinge on a BIT-EQUAL nearest-neighbour tie (an instance point that is the exact midpoint of
        # two map voxels is equidistant from both; scipy's cKDTree then returns whichever its traversal meets first)?
        # The oracle is run with cKDTree's own choice and with the HIP path's rule (strict float64 minimum, bit-equal
        # ties to the lowest index); instances whose pooled feature moves are flagged.  Everything else -- every
        # distance that differs by even one ulp -- is decided identically by both.
        from oracle import hmsg_oracle as O
        per_mode = []
        for tie in ("scipy", "exact"):
            O.NN_TIE = tie
            r = O.create_feature_map(frames, cfg)
            per_mode.append(np.stack([np.asarray(f).reshape(-1) for f in r["mask_feats"]]))
        O.NN_TIE = "scipy"
        tie_sensitive = (np.abs(per_mode[0] - per_mode[1]).max(axis=1) > 1e-7) if per_mode[0].shape == per_mode[1].shape \
            else np.ones(len(per_mode[0]), bool)
        np.savez_compressed(
            os.path.join(out_dir, name + ".npz"), **pack_frames(frames),
            cfg_keys=np.array(list(cfg.keys())), cfg_vals=np.array([str(v) for v in cfg.values()]),
            ref_cloud=cloud, ref_cloud_cols=np.asarray(g.full_pcd.colors),
            ref_full_feats=g.full_feats_array.astype(np.float32),
            ref_mask_pts=mp, ref_mask_off=moff, ref_mask_feats=feats.astype(np.float32),
            ref_tie_sensitive=tie_sensitive)
        print(name, "instances hinging on a bit-equal NN tie", int(tie_sensitive.sum()), "of", len(tie_sensitive))
        print(name, "cloud", cloud.shape, "instances", len(g.mask_pcds), "feats", feats.shape)


def gen_fusion(G, X, out_dir):
    """extract_feats_per_pixel (sam_clip_feats_extractor.py:82-191) on one frame."""
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=7, rooms_x=1, rooms_z=1, width=96, height=64, n_frames=4, n_masks=9, feat_dim=48)
    sc = SynthScene(spec)
    fr = sc.frame(2)
    fr["masks"][3] = False                       # an empty mask
    fr["masks"][4] = True                        # a full-frame mask
    state = dict()

    class MaskGen:
        def generate(self, image):
            return [dict(segmentation=fr["masks"][m], predicted_iou=1.0) for m in range(spec.n_masks)]

    X.get_img_feats = lambda img, pre, model: fr["f_g"].copy()
    X.crop_all_bounding_boxs = lambda image, masks, block_background, bbox_margin: \
        [("masked" if block_background else "crop", m) for m in range(len(masks))]
    X.get_img_feats_batch = lambda imgs, pre, model: (fr["f_masked"] if imgs[0][0] == "masked" else fr["f_crop"]).copy()
    X.cv2 = MagicMock()
    out, f_p, masks, f_g = X.extract_feats_per_pixel(fr["rgb"], MaskGen(), None, None, clip_feat_dim=spec.feat_dim,
                                                     bbox_margin=50, maskedd_weight=0.4418)
    np.savez_compressed(os.path.join(out_dir, "fusion.npz"), masks=np.packbits(fr["masks"], axis=-1),
                        shape=np.array(fr["masks"].shape), f_g=fr["f_g"], f_masked=fr["f_masked"],
                        f_crop=fr["f_crop"], ref_f2d=out.numpy(), ref_f_p=f_p.numpy())
    print("fusion", out.shape, out.dtype)


def gen_feats_dbscan(G, X, out_dir):
    """feats_denoise_dbscan (graph_utils.py:682-728) on three shapes of input."""
    from memory.hmsg.utils.graph_utils import feats_denoise_dbscan
    rng = np.random.Generator(np.random.PCG64(11))
    D = 24
    cases = {}
    base = rng.standard_normal((3, D)).astype(np.float32)
    base /= np.linalg.norm(base, axis=1, keepdims=True)
    # (a) two dense clusters + noise, (b) no cluster (too few), (c) single tight cluster with scale jitter
    a = np.concatenate([base[0] + 0.01 * rng.standard_normal((180, D)), base[1] + 0.01 * rng.standard_normal((140, D)),
                        rng.standard_normal((25, D))]).astype(np.float32)
    a = a[rng.permutation(len(a))]
    b = (base[2] + 0.3 * rng.standard_normal((40, D))).astype(np.float32)
    c = ((base[0] + 0.004 * rng.standard_normal((130, D))) * rng.uniform(0.2, 1.0, (130, 1))).astype(np.float32)
    for k, v in dict(a=a, b=b, c=c).items():
        cases["in_" + k] = v
        cases["ref_" + k] = np.asarray(feats_denoise_dbscan(v, eps=0.01, min_points=100), dtype=np.float32)
    np.savez_compressed(os.path.join(out_dir, "feats_dbscan.npz"), **cases)
    print("feats_dbscan", {k: v.shape for k, v in cases.items()})


def gen_query(G, X, out_dir):
    """query_floor / query_hmsg_room / query_hmsg_object (graph.py:2216-2257, 3056-3272) on a synthetic
    node table with a deterministic text table in place of the CLIP text encoder."""
    rng = np.random.Generator(np.random.PCG64(5))
    D, R, N = 40, 6, 90
    words = ["background", "wall"] + ["room%d" % i for i in range(R)] + ["thing%d" % i for i in range(20)] + \
            ["floor %d" % i for i in range(2)]
    table = {}
    for w in words:
        t = rng.standard_normal((2, D)).astype(np.float32)
        t /= np.linalg.norm(t, axis=1, keepdims=True)
        table[w] = t.mean(axis=0)

    def text_feats(in_text, clip_model, clip_feat_dim, batch_size=64):
        return np.stack([table[w] for w in in_text]).astype(np.float32)

    G.get_text_feats_multiple_templates = text_feats
    g = G.Graph.__new__(G.Graph)
    g.clip_model, g.clip_feat_dim = None, D
    ns = types.SimpleNamespace
    g.floors, g.rooms, g.objects = [], [], []
    for f in range(2):
        g.floors.append(ns(floor_id=str(f), floor_zero_level=3.0 * (1 - f), rooms=[]))   # order reversed on purpose
    obj_emb = np.zeros((N, D), np.float32)
    for r in range(R):
        fl = g.floors[r % 2]
        room = ns(room_id="%s_%d" % (fl.floor_id, len(fl.rooms)), name="room%d" % r, objects=[],
                  embeddings=[rng.standard_normal(D) for _ in range(3 + r)])
        fl.rooms.append(room)
        g.rooms.append(room)
    for o in range(N):
        room = g.rooms[int(rng.integers(0, R))]
        e = table["thing%d" % (o % 20)] + 0.25 * rng.standard_normal(D).astype(np.float32)
        obj_emb[o] = e
        obj = ns(object_id="%s_%d" % (room.room_id, len(room.objects)), room_id=room.room_id,
                 embedding=e.astype(np.float64), name="thing")
        room.objects.append(obj)
        g.objects.append(obj)
    out = dict(table_words=np.array(words), table=np.stack([table[w] for w in words]),
               obj_emb=np.stack([o.embedding for o in g.objects]),
               obj_room=np.array([[i for i, r in enumerate(g.rooms) if r.room_id == o.room_id][0] for o in g.objects]),
               room_floor=np.array([int(r.room_id.split("_")[0]) for r in g.rooms]),
               room_name=np.array([r.name for r in g.rooms]),
               room_view_off=np.cumsum([0] + [len(r.embeddings) for r in g.rooms]),
               room_view_emb=np.concatenate([np.stack(r.embeddings) for r in g.rooms]),
               floor_zero=np.array([f.floor_zero_level for f in g.floors]))
    # floors
    out["ref_floor_int"] = np.array([g.query_floor("1"), g.query_floor("2")])
    out["ref_floor_clip"] = np.array([g.query_floor("floor 0"), g.query_floor("floor 1")])
    res_idx, res_room, res_score, res_rooms_label, res_rooms_view, qspec = [], [], [], [], [], []
    k = 5
    for q in range(12):
        obj_q = "thing%d" % q
        room_q = "room%d" % (q % R)
        floor_id = [-1, 0, 1][q % 3]
        rl = g.query_hmsg_room(room_q, floor_id=floor_id, query_method="label")
        rv = g.query_hmsg_room(room_q, floor_id=floor_id, query_method="view_embedding")
        negs = ["background"] if q % 4 else ["background", "wall"]
        oi, ri, sc = g.query_hmsg_object(obj_q, floor_id=floor_id, room_ids=rl, top_k=k, negative_prompt=negs)
        pad = lambda a, n, v=-1: list(a) + [v] * (n - len(a))
        res_idx.append(pad(oi, k)); res_room.append(pad(ri, k)); res_score.append(pad(sc, k, np.nan))
        res_rooms_label.append(pad(rl, R)); res_rooms_view.append(pad(rv, 10))
        qspec.append([q, q % R, floor_id, len(negs)])
    # all-rooms query (view mode ids feed object search globally) and a query that IS a negative label
    oi, ri, sc = g.query_hmsg_object("wall", floor_id=-1, room_ids=list(range(R)), top_k=k,
                                     negative_prompt=["background", "wall"])
    out.update(ref_obj_idx=np.array(res_idx), ref_obj_room=np.array(res_room), ref_obj_score=np.array(res_score),
               ref_rooms_label=np.array(res_rooms_label), ref_rooms_view=np.array(res_rooms_view),
               qspec=np.array(qspec), ref_neg_idx=np.array(oi), ref_neg_room=np.array(ri), ref_neg_score=np.array(sc))
    # the driver (graph.py:3483-3591) with the LLM parse replaced by a fixed (floor, room, object) triple
    triples = {
        "go to thing3 in room2 on floor 2": ("2", "room2", "thing3"),
        "find thing7 in room5": (None, "room5", "thing7"),
        "thing11 in the Exhibition room1 on floor 1": ("1", "Exhibition room1", "thing11"),
        "thing1 somewhere in room0 on floor 0": ("floor 0", "room0", "thing1"),
    }
    table["Exhibition room1"] = table["room1"]
    G.parse_hier_query_use_prompt_insentence_parse_icra = lambda cfg, instr: triples[instr]
    g.cfg = None
    drv = []
    for instr in triples:
        fl, rooms, objs, res = g.query_hierarchy_protected_icra(instr, top_k=3, use_gpt=False)
        drv.append(dict(instruction=instr, triple=list(triples[instr]), floor=None if fl is None else fl.floor_id,
                        rooms=[r.room_id for r in rooms], objects=[o.object_id for o in objs],
                        object_index=[g.objects.index(o) for o in objs], negative_labels=res["negative_labels"]))
    import json
    out["driver_json"] = np.array(json.dumps(drv))
    np.savez_compressed(os.path.join(out_dir, "query.npz"), **out)
    print("query ok", out["ref_obj_idx"][:3])
    print("driver", drv)


def synth_building_cloud(seed, floors):
    """A voxel-grid-like cloud of a building: per storey a floor slab, a ceiling slab and four walls on a 5 cm
    lattice with sub-voxel jitter (what full_pcd looks like after voxel_down_sample + outlier removal)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    pts = []
    for y0, h, (sx, sz) in floors:
        xs, zs = np.arange(0.0, sx, 0.05), np.arange(0.0, sz, 0.05)
        gx, gz = np.meshgrid(xs, zs, indexing="ij")
        for y in (y0, y0 + h):                                           # slabs
            pts.append(np.stack([gx.ravel(), np.full(gx.size, y), gz.ravel()], 1))
        ys = np.arange(y0, y0 + h, 0.05)
        for x in (0.0, sx):                                              # walls
            gy, gz2 = np.meshgrid(ys, zs, indexing="ij")
            pts.append(np.stack([np.full(gy.size, x), gy.ravel(), gz2.ravel()], 1))
        for z in (0.0, sz):
            gy, gx2 = np.meshgrid(ys, xs, indexing="ij")
            pts.append(np.stack([gx2.ravel(), gy.ravel(), np.full(gy.size, z)], 1))
        # furniture-like clutter at mid height (keeps the histogram from being two clean spikes)
        n = 1200
        pts.append(np.stack([rng.uniform(0.3, sx - 0.3, n), rng.uniform(y0 + 0.3, y0 + 1.2, n),
                             rng.uniform(0.3, sz - 0.3, n)], 1))
    p = np.concatenate(pts) + rng.uniform(-0.02, 0.02, size=(sum(len(q) for q in pts), 3))
    return p


def gen_floors(G, X, out_dir):
    """A8: the reference's Graph.segment_floors_manually (graph.py:624-787) on full clouds."""
    o3d = sys.modules["open3d"]
    cases = {
        "one_storey": synth_building_cloud(11, [(0.0, 2.6, (3.0, 2.0))]),
        "two_storeys": synth_building_cloud(12, [(0.0, 2.7, (3.0, 2.5)), (3.0, 2.6, (3.0, 2.5))]),
        "three_storeys": synth_building_cloud(13, [(-0.2, 2.5, (2.5, 2.0)), (2.9, 2.6, (2.5, 2.0)), (6.1, 2.4, (2.5, 2.0))]),
        "tall_gap": synth_building_cloud(14, [(0.0, 2.4, (2.5, 2.0)), (5.5, 2.4, (2.5, 2.0))]),
    }
    # ... and the cloud the reference built from frames in the build fixture
    zb = np.load(os.path.join(out_dir, "build_hier.npz"), allow_pickle=True)
    cases["fixture_scene"] = np.asarray(zb["ref_cloud"], dtype=np.float64)
    out = {}
    for name, pts in cases.items():
        g = G.Graph.__new__(G.Graph)
        g.cfg = AttrDict(main=AttrDict(save_path="/tmp/hmsg_golden_tmp"),
                         pipeline=AttrDict(save_intermediate_results=False))
        g.graph_tmp_folder = "/tmp/hmsg_golden_tmp"
        g.floors = []
        pc = o3d.geometry.PointCloud()
        pc.points = pts.copy()
        g.full_pcd = pc
        ranges = g.segment_floors_manually(None)
        if name != "fixture_scene":                      # (that cloud is already stored in build_hier.npz)
            out[name + "_pts"] = pts
        out[name + "_ranges"] = np.array(ranges, dtype=np.float64).reshape(-1, 2)
        out[name + "_zero"] = np.array([f.floor_zero_level for f in g.floors], dtype=np.float64)
        out[name + "_height"] = np.array([f.floor_height for f in g.floors], dtype=np.float64)
        out[name + "_vertices"] = np.stack([np.asarray(f.vertices) for f in g.floors])
        out[name + "_npts"] = np.array([len(np.asarray(f.pcd.points)) for f in g.floors], dtype=np.int64)
        print(name, "floors", out[name + "_ranges"].tolist())
    out["cases"] = np.array(list(cases.keys()))
    np.savez_compressed(os.path.join(out_dir, "floors.npz"), **out)


def objects_case_inputs(zb):
    """Rooms + label table used by the A10 fixture (also imported by the test that replays it)."""
    cloud = np.asarray(zb["ref_cloud"], dtype=np.float64)
    lo, hi = cloud.min(0), cloud.max(0)
    xm = lo[0] + 0.55 * (hi[0] - lo[0])                                  # two rooms, split along x, 5 cm vertex grids
    rooms = []
    for x0, x1 in ((lo[0], xm), (xm, hi[0])):
        xs, zs = np.arange(x0, x1, 0.05), np.arange(lo[2], hi[2], 0.05)
        rooms.append(np.stack(np.meshgrid(xs, zs, indexing="ij"), -1).reshape(-1, 2))
    D = int(np.asarray(zb["ref_mask_feats"]).shape[1])
    rng = np.random.Generator(np.random.PCG64(2024))
    text = rng.standard_normal((9, D)).astype(np.float32)
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    classes = ["label%d" % i for i in range(9)]
    return rooms, text, classes


OBJECT_VIEW_FRAMES = [[0, 7, 14], [21, 28, 35]]       # frames whose poses become the views of the two rooms (objects_views)


def gen_objects(G, X, out_dir):
    """A10: the reference's Graph.segment_hmsg_objects (graph.py:1582-1736) on the instances the reference built in
    the build_seq fixture, with two given rooms (rooms are an input of the path): once without views (objects.npz) and
    once with three views per room taken from the fixture's camera poses (objects_views.npz: the view <-> object
    topology and best_view_id of :1706-1733)."""
    o3d = sys.modules["open3d"]
    from memory.hmsg.graph.room import Room
    from memory.hmsg.graph.view import View
    zb = np.load(os.path.join(out_dir, "build_seq.npz"), allow_pickle=True)
    rooms, text, classes = objects_case_inputs(zb)
    G.get_label_feats = lambda *a, **k: (text, classes)

    def run(with_views):
        g = G.Graph.__new__(G.Graph)
        g.cfg = AttrDict(main=AttrDict(save_path="/tmp/hmsg_golden_tmp"),
                         pipeline=AttrDict(save_intermediate_results=False, obj_labels="synthetic"))
        g.graph_tmp_folder = "/tmp/hmsg_golden_tmp"
        g.floors, g.rooms, g.objects, g.views = [], [], [], []
        g.clip_model, g.clip_feat_dim = None, text.shape[1]
        pc = o3d.geometry.PointCloud()
        pc.points = np.asarray(zb["ref_cloud"], dtype=np.float64).copy()
        g.full_pcd = pc
        g.segment_floors_manually(None)
        fl = g.floors[0]
        for k, verts in enumerate(rooms):
            r = Room("%s_%d" % (fl.floor_id, k), fl.floor_id)
            r.vertices = verts
            if with_views:
                for fidx in OBJECT_VIEW_FRAMES[k]:
                    v = View("%s_%d" % (r.room_id, len(g.views)), r.room_id, fidx)
                    r.views.append(v)
                    g.views.append(v)
            fl.add_room(r)
            g.rooms.append(r)
        off = zb["ref_mask_off"]
        g.mask_pcds = []
        for k in range(len(off) - 1):
            m = o3d.geometry.PointCloud()
            m.points = np.asarray(zb["ref_mask_pts"][off[k]:off[k + 1]], dtype=np.float64).copy()
            m.colors = np.zeros_like(m.points)
            g.mask_pcds.append(m)
        g.mask_feats = [np.asarray(f) for f in zb["ref_mask_feats"]]

        class DS:
            def get_camera_intrinsics(self):
                return np.asarray(zb["K"])

            def __getitem__(self, i):
                return np.asarray(zb["rgb"][i]), None, np.asarray(zb["pose"][i]), None, None
        g.dataset = DS()
        g.segment_hmsg_objects()
        return g
    g = run(False)
    mask_of = []
    for o in g.objects:                      # which instance became this object (the embedding is the instance's)
        mask_of.append(next(i for i, f in enumerate(g.mask_feats) if f is o.embedding or np.array_equal(f, o.embedding)))
    np.savez_compressed(
        os.path.join(out_dir, "objects.npz"),
        obj_mask=np.array(mask_of, np.int64),
        obj_id=np.array([o.object_id for o in g.objects]),
        obj_room=np.array([o.room_id for o in g.objects]),
        obj_name=np.array([o.name for o in g.objects]),
        obj_npts=np.array([len(np.asarray(o.pcd.points)) for o in g.objects], np.int64),
        denoised_npts=np.array([len(np.asarray(m.points)) for m in g.mask_pcds], np.int64),
        floor_zero=np.array([f.floor_zero_level for f in g.floors]), floor_height=np.array([f.floor_height for f in g.floors]))
    print("objects", len(g.objects), "of", len(g.mask_pcds), "instances; rooms",
          {r.room_id: len(r.objects) for r in g.rooms})
    import json
    gv = run(True)
    assert [o.object_id for o in gv.objects] == [o.object_id for o in g.objects]
    rec = dict(view_frames=OBJECT_VIEW_FRAMES,
               views=[dict(view_id=v.view_id, room_id=v.room_id, img_id=int(v.img_id), object_ids=list(v.object_ids),
                           text_discription=[str(t) for t in v.text_discription]) for v in gv.views],
               objects=[dict(object_id=o.object_id, view_ids=list(o.view_ids), best_view_id=o.best_view_id) for o in gv.objects])
    with open(os.path.join(out_dir, "objects_views.json"), "w") as f:
        json.dump(rec, f)
    print("objects_views:", sum(len(v["object_ids"]) for v in rec["views"]), "view-object links,",
          sum(o["best_view_id"] is not None for o in rec["objects"]), "objects with a best view")


def persist_case():
    """A small graph (1 floor, 2 rooms, 3 objects, 2 views) described with plain data: the A11 fixture builds it
    with the reference's classes, the test with the mirror's."""
    rng = np.random.Generator(np.random.PCG64(5))
    cloud = lambda n: rng.uniform(0.0, 2.0, size=(n, 3))
    return dict(
        floor=dict(floor_id="0", name="floor_0", pts=cloud(40), vertices=rng.uniform(0, 3, (8, 3)), height=2.6, zero=-0.02),
        rooms=[dict(room_id="0_0", name="kitchen", pts=cloud(20), vertices=rng.uniform(0, 3, (30, 2)), height=2.6, zero=-0.02,
                    embeddings=[rng.standard_normal(8), rng.standard_normal(8)], represent=[3, 7], sample=[1, 3, 7]),
               dict(room_id="0_1", name=None, pts=cloud(10), vertices=rng.uniform(0, 3, (12, 2)), height=2.6, zero=-0.02,
                    embeddings=[], represent=[], sample=[])],
        objects=[dict(object_id="0_0_0", room="0_0", name="chair", pts=cloud(15), embedding=rng.standard_normal(8), views=["0_0_0"], best="0_0_0"),
                 dict(object_id="0_0_1", room="0_0", name="table", pts=cloud(25), embedding=rng.standard_normal(8).astype(np.float32), views=[], best=None),
                 dict(object_id="0_1_0", room="0_1", name="lamp", pts=cloud(12), embedding=None, views=["0_1_1"], best="0_1_1")],
        views=[dict(view_id="0_0_0", room="0_0", img_id=np.int64(3), objects=["0_0_0"], text=["chair"], img_path="rgb/000003.png"),
               dict(view_id="0_1_1", room="0_1", img_id=7, objects=["0_1_0"], text=["lamp"], img_path=None)])


def build_persist_graph(case, Floor, Room, Object, View, make_pcd):
    """Instantiate `case` with a set of node classes (the reference's or the mirror's); returns (floor, rooms, objects, views)."""
    fl = Floor(case["floor"]["floor_id"], name=case["floor"]["name"])
    fl.pcd, fl.vertices = make_pcd(case["floor"]["pts"]), np.asarray(case["floor"]["vertices"])
    fl.floor_height, fl.floor_zero_level = case["floor"]["height"], case["floor"]["zero"]
    views = {}
    for v in case["views"]:
        o = View(v["view_id"], v["room"], v["img_id"])
        o.object_ids, o.text_discription, o.img_path = list(v["objects"]), list(v["text"]), v["img_path"]
        views[v["view_id"]] = o
    rooms = {}
    for r in case["rooms"]:
        o = Room(r["room_id"], fl.floor_id, name=r["name"])
        o.pcd, o.vertices = make_pcd(r["pts"]), np.asarray(r["vertices"])
        o.room_height, o.room_zero_level = r["height"], r["zero"]
        o.embeddings = [np.asarray(e) for e in r["embeddings"]]
        o.represent_images, o.sample_images = list(r["represent"]), list(r["sample"])
        o.views = [v for v in views.values() if v.room_id == r["room_id"]]
        fl.add_room(o)
        rooms[r["room_id"]] = o
    objects = []
    for q in case["objects"]:
        o = Object(q["object_id"], q["room"])
        o.name, o.pcd, o.embedding = q["name"], make_pcd(q["pts"]), q["embedding"]
        o.vertices = np.asarray(q["pts"])[:, [0, 2]]
        o.view_ids, o.best_view_id = list(q["views"]), q["best"]
        rooms[q["room"]].add_object(o)
        objects.append(o)
    return fl, list(rooms.values()), objects, list(views.values())


def gen_persist(G, X, out_dir):
    """A11: the JSON records the reference's own Floor / Room / Object / View .save() write (floor.py:33-51,
    room.py:309-333, object.py:37-57, view.py:62-73) for a small graph."""
    import json
    import tempfile
    o3d = sys.modules["open3d"]
    from memory.hmsg.graph.floor import Floor
    from memory.hmsg.graph.object import Object
    from memory.hmsg.graph.room import Room
    from memory.hmsg.graph.view import View

    def make_pcd(p):
        pc = o3d.geometry.PointCloud()
        pc.points = np.asarray(p, dtype=np.float64)
        return pc
    fl, rooms, objects, views = build_persist_graph(persist_case(), Floor, Room, Object, View, make_pcd)
    tmp = tempfile.mkdtemp()
    records = {}
    for kind, nodes in (("floors", [fl]), ("rooms", rooms), ("objects", objects), ("views", views)):
        d = os.path.join(tmp, kind)
        os.makedirs(d)
        for n in nodes:
            n.save(d)
        records[kind] = {f: json.load(open(os.path.join(d, f))) for f in sorted(os.listdir(d)) if f.endswith(".json")}
    json.dump(records, open(os.path.join(out_dir, "persist.json"), "w"), indent=0)
    print("persist records", {k: sorted(v) for k, v in records.items()})


def gen_roomnames(G, X, out_dir):
    """A12 helper: Room.infer_room_type_from_view_embedding (room.py:131-172) -- per view arg-max over the room-type
    text features, majority vote (np.unique order on ties), "unknown room type" without embeddings."""
    import memory.hmsg.graph.room as R
    rng = np.random.Generator(np.random.PCG64(31))
    D, types = 16, ["kitchen", "office", "bedroom", "corridor", "bathroom"]
    text = rng.standard_normal((len(types), D))
    text /= np.linalg.norm(text, axis=1, keepdims=True)
    R.get_text_feats_multiple_templates = lambda names, model, dim: text
    embs, names = [], []
    for k in (0, 1, 2, 5, 6, 9):
        e = rng.standard_normal((k, D))
        if k == 6:                      # a 3:3 tie between two types (lower type id wins through np.unique)
            e = np.concatenate([np.tile(text[3], (3, 1)), np.tile(text[1], (3, 1))]) + 0.01 * rng.standard_normal((6, D))
        room = R.Room("0_%d" % len(embs), "0")
        room.embeddings = [v for v in e]
        names.append(room.infer_room_type_from_view_embedding(types, None, D))
        embs.append(e)
    np.savez_compressed(os.path.join(out_dir, "roomnames.npz"), text=text, types=np.array(types), names=np.array(names),
                        counts=np.array([len(e) for e in embs]), embs=np.concatenate(embs))
    print("room names", names)


def mergeobj_case():
    """Objects of one room for Room.merge_objects (room.py:62-129): (name, points, embedding).  Boxes of 4 cm
    lattice points; the x offsets decide which same-name objects overlap within 10 cm."""
    rng = np.random.Generator(np.random.PCG64(77))

    def box(x0, n=6):
        g = np.stack(np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij"), -1).reshape(-1, 3) * 0.04
        return g + [x0, 0.0, 0.0] + rng.uniform(-0.005, 0.005, size=g.shape)
    specs = [("chair", box(0.00)), ("chair", box(0.15)), ("table", box(0.05)), ("chair", box(2.00)),
             ("lamp", box(3.00)), ("lamp", box(3.12)), ("lamp", box(3.24)), ("shelf", np.zeros((0, 3))),
             ("shelf", box(5.00)), ("table", box(6.00))]
    return [(n, p, rng.standard_normal(8)) for n, p in specs]


def gen_mergeobjects(G, X, out_dir):
    """N2: the reference's Room.merge_objects (same-name fusion, room.py:62-129; Object.__add__ object.py:93-103)."""
    o3d = sys.modules["open3d"]
    from memory.hmsg.graph.object import Object
    from memory.hmsg.graph.room import Room
    room = Room("0_3", "0")
    case = mergeobj_case()
    for k, (name, pts, emb) in enumerate(case):
        o = Object("0_3_%d" % k, "0_3")
        o.name = name
        pc = o3d.geometry.PointCloud()
        pc.points = pts.copy()
        o.pcd, o.embedding = pc, emb.copy()
        o.vertices = pts[:, [0, 2]].copy()
        room.add_object(o)
    room.merge_objects()
    out = dict(n=np.array(len(room.objects)), ids=np.array([o.object_id for o in room.objects]),
               names=np.array([o.name for o in room.objects]),
               npts=np.array([len(np.asarray(o.pcd.points)) for o in room.objects], np.int64),
               emb=np.stack([np.asarray(o.embedding, np.float64) for o in room.objects]),
               pts=np.concatenate([np.asarray(o.pcd.points).reshape(-1, 3) for o in room.objects]))
    for k, o in enumerate(room.objects):
        out["vertices_%d" % k] = np.asarray(o.vertices, np.float64)
    np.savez_compressed(os.path.join(out_dir, "mergeobjects.npz"), **out)
    print("merged", len(case), "->", len(room.objects), out["ids"].tolist(), out["names"].tolist(), out["npts"].tolist())


def roomemb_case():
    """Rooms (x/z lattices lifted to clouds), camera poses and per-image CLIP features for compute_room_embeddings:
    one room with more than 24 images (KMeans path), one with fewer, one that gets no camera inside the floor bounds
    (forced assignment from the cameras OUTSIDE the bounds), cameras above / below the floor."""
    rng = np.random.Generator(np.random.PCG64(2025))
    D = 20

    def room(x0, z0, w, h):
        xs, zs = np.arange(x0, x0 + w, 0.25), np.arange(z0, z0 + h, 0.25)
        g = np.stack(np.meshgrid(xs, zs, indexing="ij"), -1).reshape(-1, 2)
        return np.stack([g[:, 0], rng.uniform(0.0, 2.5, len(g)), g[:, 1]], axis=1)
    rooms = [room(0, 0, 4, 3), room(5, 0, 3, 3), room(0, 5, 2, 2), room(9, 9, 1, 1)]
    poses, embs = [], []
    centres = rng.standard_normal((6, D))

    def cam(x, y, z, c):
        T = np.eye(4)
        T[:3, 3] = [x, y, z]
        e = centres[c] + 0.15 * rng.standard_normal(D)
        poses.append(T)
        embs.append((e / np.linalg.norm(e)).astype(np.float32)[None, :])
    for k in range(40):
        cam(rng.uniform(0.2, 3.8), 1.5, rng.uniform(0.2, 2.8), k % 5)        # room 0: 40 images -> KMeans(24)
    for k in range(7):
        cam(rng.uniform(5.2, 7.8), 1.4, rng.uniform(0.2, 2.8), 5)            # room 1: 7 images
    cam(1.0, 9.0, 6.0, 2)                                                      # above the floor: never assigned normally
    cam(9.4, -3.0, 9.4, 3)                                                     # below the floor, next to room 3
    for k in range(3):
        cam(rng.uniform(0.2, 1.8), 1.6, rng.uniform(5.2, 6.8), 1)            # room 2
    return rooms, poses, embs, np.array([-0.1, 0.0, -0.1]), np.array([10.0, 2.6, 10.0])


def gen_roomemb(G, X, out_dir):
    """A9: the reference's compute_room_embeddings (utils/graph_utils.py:192-356)."""
    import tempfile
    o3d = sys.modules["open3d"]
    from memory.hmsg.utils.graph_utils import compute_room_embeddings
    rooms, poses, embs, pmin, pmax = roomemb_case()
    pcds = []
    for r in rooms:
        pc = o3d.geometry.PointCloud()
        pc.points = r
        pcds.append(pc)
    repr_embs, repr_ids, r2i, clip = compute_room_embeddings(pcds, poses, embs, pmin, pmax, 24, tempfile.mkdtemp())
    out = dict(n_rooms=np.array(len(rooms)))
    for i in range(len(rooms)):
        out["repr_ids_%d" % i] = np.array(repr_ids[i], np.int64)
        out["repr_embs_%d" % i] = np.array(repr_embs[i], np.float32).reshape(len(repr_ids[i]), -1)
        out["img_ids_%d" % i] = np.array(r2i[i], np.int64)
        out["clip_%d" % i] = np.asarray(clip[i], np.float32)
    np.savez_compressed(os.path.join(out_dir, "roomemb.npz"), **out)
    print("roomemb", {i: (len(r2i[i]), len(repr_ids[i])) for i in range(len(rooms))})


def gen_graphedges(G, X, out_dir):
    """A11: edges of the reference's create_graph_new (graph.py:1752-1775) on a freshly built graph (View.room_id is the
    int room index there, so no Room - View edge appears) and of load_hmsg_graph (:1892-1987) after saving it."""
    import json
    import tempfile
    import networkx as nx
    o3d = sys.modules["open3d"]
    from memory.hmsg.graph.floor import Floor
    from memory.hmsg.graph.object import Object
    from memory.hmsg.graph.room import Room
    from memory.hmsg.graph.view import View

    def make_pcd(p):
        pc = o3d.geometry.PointCloud()
        pc.points = np.asarray(p, dtype=np.float64)
        return pc

    def node_key(n):
        for cls, tag, attr in ((Floor, "floor", "floor_id"), (Room, "room", "room_id"), (Object, "object", "object_id"),
                               (View, "view", "view_id")):
            if isinstance(n, cls):
                return "%s:%s" % (tag, getattr(n, attr))
        return "root:%s" % n

    def edges(g):
        return sorted(sorted([node_key(a), node_key(b)]) for a, b in g.graph.edges())
    fl, rooms, objects, views = build_persist_graph(persist_case(), Floor, Room, Object, View, make_pcd)
    for v in views:                       # build time: the per-floor room INDEX (graph.py:1176-1183)
        v.room_id = int(str(v.room_id).split("_")[-1])
    g = G.Graph.__new__(G.Graph)
    g.graph = nx.Graph()
    g.floors, g.rooms, g.objects, g.views = [fl], rooms, objects, views
    g.create_graph_new()
    built = edges(g)
    tmp = tempfile.mkdtemp()
    g.save_hmsg_graph(tmp)
    g2 = G.Graph.__new__(G.Graph)
    g2.graph = nx.Graph()
    g2.floors, g2.rooms, g2.objects, g2.views = [], [], [], []
    g2.load_hmsg_graph(tmp)
    json.dump(dict(built=built, loaded=edges(g2), object_order=[o.object_id for o in g2.objects],
                   view_order=[v.view_id for v in g2.views]), open(os.path.join(out_dir, "graphedges.json"), "w"), indent=0)
    print("graph edges: built", len(built), "loaded", len(edges(g2)))


def main():
    out_dir = os.path.join(REPO, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    G, X = import_reference()
    which = sys.argv[1:] or ["fusion", "feats_dbscan", "query", "build"]
    for w in which:
        globals()["gen_" + w](G, X, out_dir)


if __name__ == "__main__":
    main()
