"""TEST INFRASTRUCTURE -- CPU restatement of the encoder-side crop / resize batching.

Reference: /root/reference/fsr_vln/memory/hmsg/utils/sam_utils.py
  increase_bbox_by_margin :58-81, crop_all_bounding_boxs :119-147, crop_image :150-164, crop_bbox :167-183;
call sites perception/models/sam_clip_feats_extractor.py:148-151 (both variants per frame, bbox_margin from the config).

Only tests/ may import this module; the product path is holoagent_amd/csrc/hmsg_crop.hip.

cv2.resize(crop, (512, 512)) is OpenCV 4.8.1 (environment.yaml:31), absent here: its INTER_LINEAR path for 8-bit images
is restated from the published source (imgproc/resize.cpp: the coefficient tables of resize(), HResizeLinear, and the
uchar specialisation of VResizeLinear) and is "parity unpinned":
  scale = 1 / (dst / src) in double; for every destination index d: f = float((d + 0.5) * scale - 0.5), s = floor(f),
  f -= s; horizontally s < 0 -> (s, f) = (0, 0) and s >= src - 1 -> (s, f) = (src - 1, 0); vertically the two source rows
  are clamped to [0, src - 1] with the weights left alone; weights are saturate_cast<short>(w * 2048) (round half to
  even); horizontal pass in int32: S[s] * a0 + S[s + 1] * a1; vertical pass:
  (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.
The slicing, margin and masking logic around it is plain numpy and is mirrored exactly (including numpy's clamping of
slices that run past the image).
"""
from __future__ import annotations

import numpy as np

COEF_SCALE = 2048


def increase_bbox_by_margin(bbox, margin):
    x, y, w, h = bbox
    x -= margin
    y -= margin
    w += margin * 2
    h += margin * 2
    if x < 0:
        w += x
        x = 0
    if y < 0:
        h += y
        y = 0
    return x, y, w, h


def _coef(dst, src):
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _short(w):
    return np.clip(np.rint(w.astype(np.float32) * np.float32(COEF_SCALE)), -32768, 32767).astype(np.int64)


def resize_linear_u8(src, dsize):
    """cv2.resize(src, dsize) for an H x W x C uint8 image, INTER_LINEAR (see the module header)."""
    dw, dh = dsize
    sh, sw = src.shape[:2]
    if sh == 0 or sw == 0:
        raise ValueError("resize of an empty image (cv2 raises: !ssize.empty())")
    sx, fx = _coef(dw, sw)
    lo, hi = sx < 0, sx >= sw - 1
    fx = np.where(lo | hi, np.float32(0), fx).astype(np.float32)
    sx = np.where(lo, 0, np.where(hi, sw - 1, sx))
    a0, a1 = _short(np.float32(1.0) - fx), _short(fx)
    sx1 = np.minimum(sx + 1, sw - 1)
    S = src.astype(np.int64)
    rows = S[:, sx] * a0[None, :, None] + S[:, sx1] * a1[None, :, None]          # [sh, dw, C] int
    sy, fy = _coef(dh, sh)
    b0, b1 = _short(np.float32(1.0) - fy), _short(fy)
    y0 = np.clip(sy, 0, sh - 1)
    y1 = np.clip(sy + 1, 0, sh - 1)
    r0, r1 = rows[y0], rows[y1]
    out = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def crop_bbox(image, bbox, bbox_margin=0):
    x, y, w, h = increase_bbox_by_margin(bbox, bbox_margin)
    x, y, w, h = int(x), int(y), int(w), int(h)
    return image[y: y + h, x: x + w]


def crop_image(image, mask):
    x, y, w, h = mask["bbox"]
    masked = image * np.expand_dims(mask["segmentation"], -1)
    x, y, w, h = int(x), int(y), int(w), int(h)
    return masked[y: y + h, x: x + w, :]


def crop_all_bounding_boxs(image, masks, block_background=False, bbox_margin=0, size=512):
    out = []
    for mask in masks:
        crop = crop_image(image, mask) if block_background else crop_bbox(image, mask["bbox"], bbox_margin)
        out.append(resize_linear_u8(crop, (size, size)))
    return out
