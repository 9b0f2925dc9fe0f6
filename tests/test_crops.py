"""Row N4: crop / resize batching (include/hmsg.h: hmsg_crop_resize_batch) against the reference's own
crop_all_bounding_boxs outputs (tests/golden/crops.npz, made by oracle/refdrive/gen_golden_crops.py) and the oracle."""
import hashlib
import os

import numpy as np
import pytest

from oracle import crop_oracle as CO
from tests import parity_common as PC

GOLD = os.path.join(os.path.dirname(__file__), "golden", "crops.npz")


def _gold_masks():
    z = np.load(GOLD)
    masks = [{"segmentation": s, "bbox": [int(v) for v in b]} for s, b in zip(z["segs"], z["bbox"])]
    return z, z["image"], masks


def _check_gold(z, margin, plain, masked):
    for name, crops in (("plain", plain), ("masked", masked)):
        for i, c in enumerate(crops):
            assert np.array_equal(c[(i % 8)::8, ((3 * i) % 8)::8], z[f"m{margin}_{name}_sub"][i]), (name, i)
            assert hashlib.sha1(np.ascontiguousarray(c).tobytes()).hexdigest() == str(z[f"m{margin}_{name}_sha1"][i]), (name, i)


@pytest.mark.parametrize("margin", [0, 7, 50])
def test_oracle_equals_reference_run(margin):
    z, image, masks = _gold_masks()
    _check_gold(z, margin, CO.crop_all_bounding_boxs(image, masks, False, margin), CO.crop_all_bounding_boxs(image, masks, True, margin))


def test_oracle_resize_properties():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    assert np.array_equal(CO.resize_linear_u8(img, (53, 37)), img)                    # same size: identity
    flat = np.full((5, 9, 3), 201, np.uint8)
    assert (CO.resize_linear_u8(flat, (512, 512)) == 201).all()                       # weights sum to one
    one = CO.resize_linear_u8(img[:1, :1], (8, 8))
    assert (one == img[0, 0]).all()                                                   # 1 x 1 source
    with pytest.raises(ValueError):
        CO.resize_linear_u8(img[:0], (8, 8))


def _random_frame(seed, H, W, M):
    rng = np.random.default_rng(seed)
    image = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    masks = []
    for m in range(M):
        w, h = int(rng.integers(1, W // 2)), int(rng.integers(1, H // 2))
        x, y = int(rng.integers(0, W - w)), int(rng.integers(0, H - h))
        seg = np.zeros((H, W), bool)
        seg[y:y + h + 1, x:x + w + 1] = rng.random((h + 1, w + 1)) < 0.6
        masks.append({"segmentation": seg, "bbox": [x, y, w, h]})
    return image, masks


def check_lib(L, big):
    from holoagent_amd._lib import HmsgError, crop_all_bounding_boxs
    z, image, masks = _gold_masks()
    for margin in (0, 7, 50):
        plain, masked = crop_all_bounding_boxs(image, masks, margin, lib_=L)
        _check_gold(z, margin, plain, masked)
    H, W, M, S = big
    image, masks = _random_frame(3, H, W, M)
    plain, masked = crop_all_bounding_boxs(image, masks, 13, size=S, lib_=L)
    for m in range(M):
        assert np.array_equal(plain[m], CO.resize_linear_u8(CO.crop_bbox(image, masks[m]["bbox"], 13), (S, S)))
        assert np.array_equal(masked[m], CO.resize_linear_u8(CO.crop_image(image, masks[m]), (S, S)))
    only_plain, none = crop_all_bounding_boxs(image, masks, 13, size=S, masked=False, lib_=L)
    assert none is None and np.array_equal(only_plain, plain)
    bad = [dict(masks[0], bbox=[5, 5, 0, 10])]                # SAM reports w = 0 for a one-column mask: cv2.resize raises
    with pytest.raises(HmsgError):
        crop_all_bounding_boxs(image, bad, 0, size=S, lib_=L)
    ok, _ = crop_all_bounding_boxs(image, bad, 4, size=S, masked=False, lib_=L)      # (the margin makes the plain crop valid)
    assert np.array_equal(ok[0], CO.resize_linear_u8(CO.crop_bbox(image, bad[0]["bbox"], 4), (S, S)))


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_simulator_equals_reference_run_and_oracle():
    from holoagent_amd._lib import HmsgLib
    check_lib(HmsgLib(PC.EMU_PATH), (48, 64, 5, 64))


@pytest.mark.gpu
def test_gpu_equals_reference_run_and_oracle():
    from holoagent_amd._lib import lib
    check_lib(lib(), (480, 640, 32, 512))


@pytest.mark.gpu
def test_gpu_device_pointers():
    """image, masks and outputs resident in HBM (how the encoders consume them): same bytes as the host-pointer call"""
    import torch

    from holoagent_amd._lib import _ptr, crop_all_bounding_boxs, lib
    L = lib()
    image, masks = _random_frame(5, 480, 640, 32)
    plain, masked = crop_all_bounding_boxs(image, masks, 50)
    dev = torch.device("cuda:0")
    t_img = torch.from_numpy(image).to(dev)
    t_seg = torch.from_numpy(np.stack([m["segmentation"] for m in masks]).astype(np.uint8)).to(dev)
    t_plain = torch.zeros((32, 512, 512, 3), dtype=torch.uint8, device=dev)
    t_masked = torch.zeros_like(t_plain)
    bbox = np.ascontiguousarray([m["bbox"] for m in masks], dtype=np.float64)
    torch.cuda.synchronize()
    rc = L.c.hmsg_crop_resize_batch(0, 480, 640, _ptr(t_img), 32, _ptr(t_seg), _ptr(bbox), 50.0, 512, _ptr(t_plain), _ptr(t_masked), None)
    assert rc == 0
    assert np.array_equal(t_plain.cpu().numpy(), plain) and np.array_equal(t_masked.cpu().numpy(), masked)
