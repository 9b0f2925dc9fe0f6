"""Golden vectors of the label vocabulary loader: runs the REFERENCE's own get_label_feats
(memory/hmsg/utils/label_feats.py:38-126, imported from /root/reference in the build container) for every CSV
vocabulary with a stand-in text encoder and stores count / first / last / SHA-1 of the class lists in
tests/golden/labels.json (the lists themselves are the reference's data and stay there).

    python -m oracle.refdrive.gen_golden_labels
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)


def digest(classes):
    return hashlib.sha1("\n".join(str(c) for c in classes).encode()).hexdigest()


def main():
    from oracle.refdrive.gen_golden import REF, import_reference
    import_reference()
    import memory.hmsg.utils.label_feats as LF
    seen = {}

    def fake_encode(classes, clip_model, dim):
        seen["n"] = len(classes)
        return np.zeros((len(classes), dim), np.float32)
    LF.get_text_feats_multiple_templates = fake_encode
    real_save, cwd = np.save, os.getcwd()
    np.save = lambda *a, **k: None                  # the reference would write its cache into its own checkout
    os.chdir(REF)                                   # (it reads "memory/hmsg/labels" relative to the working directory)
    out = {}
    try:
        for name in ("HM3DSEM_LABELS", "IMAGENET21K_LABELS", "SCANNET200", "SCANNET20", "FINALLABEL"):
            feats, classes = LF.get_label_feats(None, 8, name)
            assert feats.shape == (len(classes), 8) and seen["n"] == len(classes)
            out[name] = {"n": len(classes), "first": str(classes[0]), "last": str(classes[-1]), "sha1": digest(classes)}
            print(name, out[name])
    finally:
        np.save = real_save
        os.chdir(cwd)
    with open(os.path.join(REPO, "tests", "golden", "labels.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
