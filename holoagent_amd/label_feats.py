"""Label vocabulary of the object level (identify_object's text table).

Mirrors memory/hmsg/utils/label_feats.py: compute_label_feats :11-35 (load `<dir>/<cache>.npy` if it exists, else encode
the classes with the 2-template text encoder and save) and get_label_feats :38-126 (vocabulary by name).  The
vocabularies themselves are DATA of the reference checkout and are not shipped here:
  * CSV vocabularies (HM3DSEM_LABELS, IMAGENET21K_LABELS, SCANNET200, SCANNET20, FINALLABEL) are read from `label_dir`
    (the reference's `memory/hmsg/labels`) exactly like the reference reads them: `pd.read_csv(header=0, sep=";")`,
    first column -- so the FIRST LINE of the file is taken as the header and is not a class (scannet200.csv: 206
    lines -> 205 classes, "shower head" is lost).  That quirk is kept: cached `text_feats_*.npy` files made by the
    reference have exactly those rows;
  * constant vocabularies (COCO_STUFF_CLASSES, MATTERPORT_*; memory/hmsg/utils/constants.py) are registered by the
    caller with `register_label_set(name, classes)`, or passed directly as a list.
"""
from __future__ import annotations

import os
from typing import Callable, Sequence

import numpy as np

CSV_SETS = {                     # name -> (csv file, cache file)   label_feats.py:68-125
    "HM3DSEM_LABELS": ("HM3D_CountsOfObjectTypes.csv", "text_feats_HM3DSEM_LABELS.npy"),
    "IMAGENET21K_LABELS": ("imagenet21k.csv", "text_feats_IMAGENET21K_LABELS.npy"),
    "SCANNET200": ("scannet200.csv", "text_feats_SCANNET200_LABELS.npy"),
    "SCANNET20": ("scannet20.csv", "text_feats_SCANNET20_LABELS.npy"),
    "FINALLABEL": ("final_label.csv", "text_feats_FINALLABEL_LABELS.npy"),
}
_REGISTERED: dict[str, list[str]] = {}


def register_label_set(name: str, classes: Sequence[str]) -> None:
    """Constant vocabularies of the reference (constants.py) are installed by name; cache file text_feats_<name>.npy."""
    _REGISTERED[name] = list(classes)


def read_label_csv(path: str) -> list:
    """First column of a ';'-separated file whose first line is consumed as the header (pd.read_csv(header=0, sep=';');
    `classes_matrix[classes_matrix.keys()[0]].values`)."""
    import pandas as pd
    m = pd.read_csv(path, header=0, sep=";")
    return list(m[m.keys()[0]].values)


def compute_label_feats(encode: Callable[[Sequence[str]], np.ndarray], label_feat_path: str, classes: Sequence[str],
                        pre_computed_feats_path: str = "text_feats.npy"):
    """label_feats.py:11-35.  `encode(classes)` = get_text_feats_multiple_templates (clip_utils.py:257-349)."""
    cache = os.path.join(label_feat_path, pre_computed_feats_path)
    if os.path.exists(cache):
        text_feats = np.load(cache)
    else:
        text_feats = np.asarray(encode(list(classes)))
        np.save(cache, text_feats)
    return text_feats, classes


def get_label_feats(encode: Callable[[Sequence[str]], np.ndarray], obj_labels, label_dir: str | None = None):
    """label_feats.py:38-126: (text features [n, D], class names)."""
    if not isinstance(obj_labels, str):
        classes = list(obj_labels)
        return np.asarray(encode(classes)), classes
    if obj_labels in CSV_SETS:
        if label_dir is None:
            raise ValueError(f"{obj_labels}: pass label_dir (the reference's memory/hmsg/labels directory)")
        csv, cache = CSV_SETS[obj_labels]
        return compute_label_feats(encode, label_dir, read_label_csv(os.path.join(label_dir, csv)), cache)
    if obj_labels in _REGISTERED:
        path = label_dir if label_dir is not None else os.getcwd()
        return compute_label_feats(encode, path, _REGISTERED[obj_labels], f"text_feats_{obj_labels}.npy")
    raise KeyError(f"unknown label set {obj_labels!r}: register_label_set(name, classes) first, or pass a list of classes")
