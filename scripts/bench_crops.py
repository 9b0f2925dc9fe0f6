"""Row N4 micro-benchmark: both crop sets of one 640x480 frame with 32 masks -> 64 crops of 512x512x3 on cuda:0, inputs
and outputs resident in HBM.  One JSON line; the roofline is the HBM write of the crops."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch  # noqa: E402

from holoagent_amd._lib import _ptr, lib  # noqa: E402
from oracle import crop_oracle as CO  # noqa: E402
from tests.test_crops import _random_frame  # noqa: E402


def main():
    H, W, M, S = 480, 640, 32, 512
    L = lib()
    image, masks = _random_frame(5, H, W, M)
    dev = torch.device("cuda:0")
    t_img = torch.from_numpy(image).to(dev)
    t_seg = torch.from_numpy(np.stack([m["segmentation"] for m in masks]).astype(np.uint8)).to(dev)
    t_plain = torch.zeros((M, S, S, 3), dtype=torch.uint8, device=dev)
    t_masked = torch.zeros_like(t_plain)
    bbox = np.ascontiguousarray([m["bbox"] for m in masks], dtype=np.float64)
    torch.cuda.synchronize()
    ms = C.c_double(0)
    best = 1e9
    for _ in range(20):
        rc = L.c.hmsg_crop_resize_batch(0, H, W, _ptr(t_img), M, _ptr(t_seg), _ptr(bbox), 50.0, S, _ptr(t_plain), _ptr(t_masked), C.byref(ms))
        assert rc == 0
        best = min(best, ms.value)
    out_bytes = 2 * M * S * S * 3
    t0 = time.perf_counter()
    for m in masks[:4]:
        CO.resize_linear_u8(CO.crop_bbox(image, m["bbox"], 50), (S, S))
        CO.resize_linear_u8(CO.crop_image(image, m), (S, S))
    cpu = (time.perf_counter() - t0) / 4 * M
    print(json.dumps({"metric": "crop_resize_frames_per_s", "value": 1e3 / best, "unit": "frames/s", "masks": M, "crop": S,
                      "device_ms_per_frame": best, "roofline": {"bound": "hbm", "achieved": out_bytes / (best / 1e3) / 1e9,
                                                                "peak": 8000.0, "unit": "GB/s", "frac": out_bytes / (best / 1e3) / 8e12},
                      "cpu_oracle_ms_per_frame": cpu * 1e3}))


if __name__ == "__main__":
    main()
