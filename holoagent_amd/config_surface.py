"""The reference's config surface for this path (fsr_vln/config/semantic_scene_reconstruction_*.yaml, read by
fsr_vln/memory/hmsg/graph/graph.py through `self.cfg.pipeline.<key>` / `cfg.main.<key>` / `cfg.models.<...>`): what each key of
the `pipeline` section means HERE.  `check_config(cfg)` is called by `Graph(cfg)` in build mode: a key this path does not know is
an error (a typo must not silently build a different map), a value it cannot honour is an error, and every key the reference's
own configs carry is accounted for -- honoured, handed to the collaborators, or ignored exactly as the reference ignores it."""
from __future__ import annotations

HONOURED = "honoured"
COLLABORATORS = "handed to the collaborators (dataset reader / encoders: outside this path)"
IGNORED_LIKE_REFERENCE = "accepted and ignored, as in the reference (no code of fsr_vln reads it)"
IGNORED_VISUAL = "accepted and ignored: the reference's debug plots and intermediate dumps (visualisation is out of scope)"
EXTENSION = "this implementation's own switch (not in the reference's configs)"

# pipeline.<key> -> (status, where the reference reads it / what happens here)
PIPELINE = {
    "voxel_size": (HONOURED, "graph.py:348 voxel_down_sample, generic.py:186-188, 1.5 * voxel_size overlap radius -> hmsg_config.voxel_size"),
    "skip_frames": (HONOURED, "graph.py:320-338 every skip-th frame -> frame ids of create_feature_map / hmsg_graph_params.skip_frames"),
    "init_overlap_thresh": (HONOURED, "graph_utils.py:1015-1038 -> hmsg_config.init_overlap_thresh"),
    "overlap_thresh_factor": (HONOURED, "graph_utils.py:1001-1003 (hierarchical levels) -> hmsg_config.overlap_thresh_factor"),
    "iou_thresh": (HONOURED, "graph_utils.py:937-941 -> hmsg_config.iou_thresh"),
    "clip_masked_weight": (HONOURED, "sam_clip_feats_extractor.py:159-166 -> hmsg_config.clip_masked_weight"),
    "max_mask_distance": (HONOURED, "generic.py:126-127 -> hmsg_config.max_mask_distance"),
    "grid_resolution": (HONOURED, "graph.py:942-1084 room segmentation grid -> Graph.segment_hmsg_room / hmsg_segment_rooms"),
    "merge_type": (HONOURED, "graph.py:425-443 'sequential' | 'hierarchical' -> hmsg_config.merge_type"),
    "obj_labels": (HONOURED, "graph.py:1582-1600 label vocabulary -> holoagent_amd.label_feats.get_label_feats"),
    "merge_objects_graph": (HONOURED, "graph.py:2053-2058 Room.merge_objects -> hmsg_graph_params.merge_objects_graph / Room.merge_objects"),
    "clip_bbox_margin": (COLLABORATORS, "sam_clip_feats_extractor.py:148-151 crop margin of the encoder side (hmsg_crop_resize_batch takes it as an argument)"),
    "save_intermediate_results": (IGNORED_VISUAL, "graph.py:530-1246 matplotlib figures of the floor / room segmentation"),
    "feature_dbscan_eps": (IGNORED_LIKE_REFERENCE, "in every shipped yaml; feats_denoise_dbscan is called with its own eps 0.01 (graph.py:470)"),
    "min_pcd_points": (IGNORED_LIKE_REFERENCE, "in every shipped yaml; the small-cloud drop uses the literal 100 (graph.py:445-448)"),
    "depth_weighting": (IGNORED_LIKE_REFERENCE, "in every shipped yaml; no reader in fsr_vln"),
    # this implementation's own
    "max_masks": (EXTENSION, "mask slots per frame (hmsg_config.max_masks, <= 256)"),
    "views_on_device": (EXTENSION, "the view <-> object test of graph.py:1712-1734 by hmsg_object_views"),
    "room_level_beside_fusion": (EXTENSION, "start the room level before the fusion has finished"),
    "label_dir": (EXTENSION, "directory of the label csv files (the reference hard-codes its own tree)"),
}
MAIN = {"device", "use_gpt", "dataset", "scene_id", "dataset_path", "depth_cut", "save_path", "graph_path", "device_id", "split", "package_path",
        "pcd_path", "feats_path", "mask_path"}


def _items(node):
    if node is None:
        return []
    if isinstance(node, dict):
        return list(node.items())
    try:                                    # OmegaConf DictConfig / SimpleNamespace
        return list(node.items())
    except Exception:
        return list(vars(node).items())


def check_config(cfg, strict=True):
    """-> {"pipeline.<key>": status}; raises ValueError for a pipeline key this path does not know and for a value it cannot honour."""
    pipe = cfg.get("pipeline") if isinstance(cfg, dict) else getattr(cfg, "pipeline", None)
    report = {}
    for k, v in _items(pipe):
        if k not in PIPELINE:
            if strict:
                raise ValueError("pipeline.%s: unknown key (known: %s) -- holoagent_amd/config_surface.py lists what every key of the "
                                 "reference's configs means here" % (k, ", ".join(sorted(PIPELINE))))
            report["pipeline." + k] = "unknown"
            continue
        report["pipeline." + k] = PIPELINE[k][0]
        if k == "merge_type" and str(v) not in ("sequential", "hierarchical"):
            raise ValueError("pipeline.merge_type = %r: 'sequential' or 'hierarchical' (graph.py:425-443)" % (v,))
        if k in ("voxel_size", "grid_resolution") and not float(v) > 0:
            raise ValueError("pipeline.%s = %r: must be positive" % (k, v))
        if k == "skip_frames" and int(v) < 1:
            raise ValueError("pipeline.skip_frames = %r: must be >= 1" % (v,))
        if k == "iou_thresh" and float(v) < 0:
            raise ValueError("pipeline.iou_thresh = %r: must be >= 0 (the pair filter of graph_utils.py:937-941 relies on it)" % (v,))
        if k == "max_masks" and not 1 <= int(v) <= 256:
            raise ValueError("pipeline.max_masks = %r: 1 .. 256" % (v,))
    return report
