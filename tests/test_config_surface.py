"""The reference's config surface and import paths.
  * tests/golden/config_keys.json holds the KEYS (and the pipeline section's scalar values) of every config the reference ships
    (fsr_vln/config/*.yaml; made by oracle/refdrive/gen_golden_config_keys.py -- data, not the files).  Every pipeline.* key is
    accounted for by holoagent_amd/config_surface.py; the keys the path reads (semantic_scene_reconstruction.py:109-127 builds
    Graph(cfg) from them) reach hmsg_config / hmsg_graph_params with the values given; an unknown key or a value the path cannot
    honour is refused loudly.
  * the reference's callers import the graph as memory.hmsg.graph.graph (applications) and as hmsg.graph.graph
    (goal_pose_publisher.py:39): with holoagent_amd/compat on sys.path both resolve to holoagent_amd.graph."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import parity_common as PC

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
KEYS = json.load(open(os.path.join(ROOT, "tests", "golden", "config_keys.json")))
BUILD_CONFIGS = {k: v for k, v in KEYS.items() if v.get("pipeline")}


def test_every_key_of_the_reference_configs_is_accounted_for():
    from holoagent_amd.config_surface import PIPELINE, check_config, HONOURED
    assert len(BUILD_CONFIGS) >= 6
    seen = set()
    for name, cfg in BUILD_CONFIGS.items():
        rep = check_config(dict(pipeline=cfg["pipeline"]))
        assert set(rep) == {"pipeline." + k for k in cfg["pipeline"]}, name
        assert "unknown" not in rep.values(), name
        seen |= set(cfg["pipeline"])
    # the keys the reference's graph.py reads (grep `cfg.pipeline.` there) are all honoured or the collaborators' -- none dropped
    for k in ("voxel_size", "skip_frames", "init_overlap_thresh", "overlap_thresh_factor", "iou_thresh", "clip_masked_weight", "max_mask_distance",
              "grid_resolution", "merge_type", "obj_labels", "merge_objects_graph"):
        assert k in seen and PIPELINE[k][0] == HONOURED, k


def test_unknown_keys_and_impossible_values_are_refused():
    from holoagent_amd.config_surface import check_config
    base = dict(next(iter(BUILD_CONFIGS.values()))["pipeline"])
    with pytest.raises(ValueError, match="voxel_sise"):
        check_config(dict(pipeline=dict(base, voxel_sise=0.05)))
    for bad in (dict(merge_type="tree"), dict(voxel_size=0), dict(skip_frames=0), dict(iou_thresh=-0.1), dict(max_masks=1000)):
        with pytest.raises(ValueError):
            check_config(dict(pipeline=dict(base, **bad)))


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_the_values_reach_the_library(tmp_path):
    """Graph(cfg) with the reference's own keys (one value of each changed, so that a default cannot pass for it) -> the handle's
    hmsg_config carries them after create_feature_map; pipeline.merge_objects_graph reaches Room.merge_objects."""
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.graph import Graph
    from holoagent_amd.synth import SceneSpec, SynthScene
    name, ref = sorted(BUILD_CONFIGS.items())[0]
    pipe = dict(ref["pipeline"])
    pipe.update(voxel_size=0.06, skip_frames=2, init_overlap_thresh=0.7, overlap_thresh_factor=0.03, iou_thresh=0.04, clip_masked_weight=0.5,
                max_mask_distance=9.5, merge_type="hierarchical", obj_labels=["chair", "table"], merge_objects_graph=True)
    spec = SceneSpec(seed=2, rooms_x=1, rooms_z=1, room_size=(3.0, 2.4, 3.0), objects_per_room=2, width=64, height=48, n_frames=8, n_masks=6, feat_dim=16)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]

    class DS:
        frameId2imgPath = ["%d.png" % i for i in range(len(frames))]

        def __len__(self):
            return len(frames)

        def __getitem__(self, i):
            f = frames[i]
            return f["rgb"], f["depth"], np.asarray(f["pose"], np.float64).reshape(4, 4), None, f["K"]

        def get_camera_intrinsics(self):
            return frames[0]["K"]

    class Enc:
        def __init__(self):
            self.i = 0

        def extract(self, rgb):
            f = frames[self.i * 2 % len(frames)]        # (skip_frames = 2: the frames asked for are 0, 2, 4, 6)
            self.i += 1
            return dict(masks=f["masks"], f_g=f["f_g"][None], f_masked=f["f_masked"], f_crop=f["f_crop"])

        def encode_text(self, words):
            rng = np.random.Generator(np.random.PCG64(len(words)))
            v = rng.standard_normal((len(words), spec.feat_dim)).astype(np.float32)
            return v / np.linalg.norm(v, axis=1, keepdims=True)
    cfg = dict(main=dict(device_id=0, dataset="synthetic", save_path=str(tmp_path)), models=dict(clip=dict(type="ViT-B/32", feat_dim=spec.feat_dim)), pipeline=pipe)
    g = Graph(cfg, dataset=DS(), encoders=Enc(), lib=HmsgLib(PC.EMU_PATH))
    assert g.config_report["pipeline.voxel_size"] == "honoured"
    g.create_feature_map()
    c = g.scene.cfg
    assert (c.voxel_size, c.init_overlap_thresh, c.overlap_thresh_factor, c.iou_thresh, c.clip_masked_weight, c.max_mask_distance, c.merge_type) == \
        (0.06, 0.7, 0.03, 0.04, 0.5, 9.5, 1)
    assert g.scene.cfg.max_frames == 4                                  # skip_frames = 2 of 8
    g.scene.close()


def test_the_reference_import_paths_resolve(tmp_path):
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "from memory.hmsg.graph.graph import Graph as G1\n"
            "from hmsg.graph.graph import Graph as G2\n"
            "from memory.hmsg.graph.room import Room\n"
            "from hmsg.graph.object import Object\n"
            "from memory.hmsg.utils.label_feats import get_label_feats\n"
            "from memory.hmsg.utils.sam_utils import crop_all_bounding_boxs\n"
            "import holoagent_amd.graph as H\n"
            "assert G1 is H.Graph and G2 is H.Graph and Room is H.Room and Object is H.Object\n"
            "print('ok')\n") % (ROOT, os.path.join(ROOT, "holoagent_amd", "compat"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
