"""Golden vectors of row N4 (crop / resize batching): runs the REFERENCE's own crop_all_bounding_boxs
(memory/hmsg/utils/sam_utils.py:119-183, imported from /root/reference in the build container) on a seeded image and
mask set and stores inputs + outputs in tests/golden/crops.npz.

    python -m oracle.refdrive.gen_golden_crops

cv2 is absent here: cv2.resize is the restatement of oracle/crop_oracle.py (unpinned); what this fixture pins is the
reference's own slicing / margin / masking logic around it.  The 512 x 512 crops are stored sub-sampled (every 8th
pixel, shifted per crop) plus a SHA-1 of each full crop.
"""
from __future__ import annotations

import hashlib
import importlib.util
import os
import sys
import types
from unittest.mock import MagicMock

import numpy as np

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, REPO)
SCRIPT = "/root/reference/fsr_vln/memory/hmsg/utils/sam_utils.py"


def import_reference():
    from oracle import crop_oracle as CO
    cv2 = types.ModuleType("cv2")
    cv2.resize = lambda img, dsize: CO.resize_linear_u8(np.asarray(img), dsize)
    sys.modules["cv2"] = cv2
    sys.modules.setdefault("matplotlib", MagicMock())
    sys.modules.setdefault("matplotlib.pyplot", MagicMock())
    spec = importlib.util.spec_from_file_location("ref_sam_utils", SCRIPT)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def synth(seed=9, H=90, W=120):
    rng = np.random.Generator(np.random.PCG64(seed))
    yy, xx = np.mgrid[0:H, 0:W]
    image = np.stack([(xx * 2 + yy) % 256, (yy * 3 + 40) % 256, (xx + yy * yy // 7) % 256], axis=-1).astype(np.uint8)
    image ^= rng.integers(0, 32, image.shape, dtype=np.uint8)
    boxes = [(10, 12, 40, 30), (0, 0, 25, 18), (100, 70, 19, 19), (55, 5, 3, 70), (30, 60, 80, 1), (2, 80, 117, 9),
             (60, 40, 1, 1)]                       # XYWH as SAM reports them; one touches every border, thin ones too
    masks = []
    for (x, y, w, h) in boxes:
        seg = np.zeros((H, W), bool)
        sub = rng.random((h, w)) < 0.7
        sub[0, 0] = sub[-1, -1] = True
        seg[y:y + h, x:x + w] = sub
        masks.append({"segmentation": seg, "bbox": [x, y, w, h]})
    return image, masks


def subsample(c, i):
    return c[(i % 8)::8, ((3 * i) % 8)::8]


def main():
    ref = import_reference()
    image, masks = synth()
    out = {"image": image, "segs": np.stack([m["segmentation"] for m in masks]),
           "bbox": np.array([m["bbox"] for m in masks], np.float64)}
    for margin in (0, 7, 50):
        plain = ref.crop_all_bounding_boxs(image, masks, block_background=False, bbox_margin=margin)
        masked = ref.crop_all_bounding_boxs(image, masks, block_background=True, bbox_margin=margin)
        for name, crops in (("plain", plain), ("masked", masked)):
            assert all(c.shape == (512, 512, 3) and c.dtype == np.uint8 for c in crops)
            out[f"m{margin}_{name}_sub"] = np.stack([subsample(c, i) for i, c in enumerate(crops)])
            out[f"m{margin}_{name}_sha1"] = np.array([hashlib.sha1(np.ascontiguousarray(c).tobytes()).hexdigest() for c in crops])
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "crops.npz"), **out)
    print("crops.npz written:", len(masks), "masks")


if __name__ == "__main__":
    main()
