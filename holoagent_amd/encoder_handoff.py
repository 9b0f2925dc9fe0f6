"""The encoder hand-off (north_star: "PyTorch-ROCm only for the wrapped vision / CLIP encoders"): what
perception/models/sam_clip_feats_extractor.py:117-158 does per frame around the CLIP image tower -- global feature of the
frame, features of the 2 M crops (masked and plain, utils/sam_utils.py:119-181 -> hmsg_crop_resize_batch on the device) --
with a random-init ViT-B/32-SHAPED PyTorch module (clip_utils.py:63-94 wraps open_clip's ViT-B/32: patch 32, width 768,
12 layers, 12 heads, 512-d projection; open_clip and its weights are not in this image), and the tensors going to the
library BY DEVICE POINTER (`tensor.data_ptr()`), never through the host.

This is a measurement aid (bench.py reports it beside `value`, never as `value`) and the demonstration that the boundary
takes a live PyTorch-ROCm module's outputs as they are: float32, contiguous, on the device."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np


def make_vit_b32(torch, dim_out=512, width=768, layers=12, heads=12, patch=32, image=224):
    nn = torch.nn

    class Block(nn.Module):
        def __init__(self):
            super().__init__()
            self.ln1, self.ln2 = nn.LayerNorm(width), nn.LayerNorm(width)
            self.attn = nn.MultiheadAttention(width, heads, batch_first=True)
            self.mlp = nn.Sequential(nn.Linear(width, 4 * width), nn.GELU(), nn.Linear(4 * width, width))

        def forward(self, x):
            h = self.ln1(x)
            x = x + self.attn(h, h, h, need_weights=False)[0]
            return x + self.mlp(self.ln2(x))

    class ViT(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv = nn.Conv2d(3, width, patch, patch, bias=False)
            self.cls = nn.Parameter(torch.randn(width) * width ** -0.5)
            self.pos = nn.Parameter(torch.randn((image // patch) ** 2 + 1, width) * width ** -0.5)
            self.pre, self.post = nn.LayerNorm(width), nn.LayerNorm(width)
            self.blocks = nn.Sequential(*[Block() for _ in range(layers)])
            self.proj = nn.Parameter(torch.randn(width, dim_out) * width ** -0.5)

        def forward(self, x):                                  # x: [B, 3, image, image]
            x = self.conv(x).flatten(2).transpose(1, 2)
            x = torch.cat([self.cls.expand(x.shape[0], 1, -1), x], 1) + self.pos
            x = self.blocks(self.pre(x))
            return self.post(x[:, 0]) @ self.proj
    return ViT()


def measure(L, scene, inp, device, torch, frames=8, crop=224, feat_dim=512):
    """`frames` frames of the resident stream through crop -> encoder -> hmsg_add_frame_features, everything on the device.
    scene: a Scene with those frames' geometry added and the map final; inp: bench.py's resident inputs.
    Returns a dict of per-frame times and the pointer check."""
    from ._lib import _ptr
    F = min(frames, inp["masks"].shape[0])
    M, H, W = inp["masks"].shape[1:]
    torch.manual_seed(0)
    enc = make_vit_b32(torch, dim_out=feat_dim, image=crop).to(device).half().eval()
    mean = torch.tensor([0.48145466, 0.4578275, 0.40821073], device=device).view(1, 3, 1, 1)     # clip preprocess constants
    std = torch.tensor([0.26862954, 0.26130258, 0.27577711], device=device).view(1, 3, 1, 1)

    def embed(u8_nhwc):
        x = (u8_nhwc.permute(0, 3, 1, 2).float() / 255.0 - mean) / std
        with torch.no_grad():
            f = enc(x.half()).float()
        return torch.nn.functional.normalize(f, dim=-1).contiguous()
    plain = torch.empty((M, crop, crop, 3), dtype=torch.uint8, device=device)
    masked = torch.empty((M, crop, crop, 3), dtype=torch.uint8, device=device)
    t_crop = t_enc = t_hand = 0.0
    on_device = True
    for it in range(-1, F):                                    # (one warm-up pass of frame 0 that is not handed over)
        f = max(it, 0)
        seg = inp["masks"][f]
        # SAM's bbox records (XYWH) of the frame's masks: a few dozen numbers, made on the device, read by the host side of the
        # crop call like the reference's `mask["bbox"]`
        ys = seg.any(dim=2)
        xs = seg.any(dim=1)
        x0 = xs.float().argmax(1)
        x1 = W - 1 - xs.flip(1).float().argmax(1)
        y0 = ys.float().argmax(1)
        y1 = H - 1 - ys.flip(1).float().argmax(1)
        some = xs.any(1)
        bbox = torch.stack([x0, y0, (x1 - x0 + 1).clamp(min=1), (y1 - y0 + 1).clamp(min=1)], 1).double()
        bbox[~some] = torch.tensor([0.0, 0.0, 8.0, 8.0], dtype=torch.float64, device=device)
        bbox_h = np.ascontiguousarray(bbox.cpu().numpy())
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        rc = L.c.hmsg_crop_resize_batch(device.index or 0, H, W, _ptr(inp["rgb"][f]), M, _ptr(seg), _ptr(bbox_h), 50.0, crop, _ptr(plain),
                                        _ptr(masked), None)
        assert rc == 0, "hmsg_crop_resize_batch failed (%d)" % rc
        t1 = time.perf_counter()
        whole = torch.nn.functional.interpolate(inp["rgb"][f].permute(2, 0, 1)[None].float(), size=(crop, crop), mode="bilinear",
                                                align_corners=False).clamp(0, 255).permute(0, 2, 3, 1).to(torch.uint8)
        f_g = embed(whole)                                      # [1, D]
        f_masked = embed(masked)                                # [M, D]
        f_crop = embed(plain)
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        if it < 0:
            continue
        for t in (f_g, f_masked, f_crop, seg):
            on_device = on_device and t.is_cuda and t.is_contiguous()
        scene.add_frame_features(f, seg[None], f_g, f_masked[None], f_crop[None])
        t3 = time.perf_counter()
        t_crop += t1 - t0
        t_enc += t2 - t1
        t_hand += t3 - t2
    return dict(frames=F, masks=M, crop=crop, encoder="random-init ViT-B/32 shape (12 x 768, patch 32, fp16), PyTorch-ROCm",
                crops_ms_per_frame=round(t_crop / F * 1e3, 3), encoder_ms_per_frame=round(t_enc / F * 1e3, 3),
                handoff_ms_per_frame=round(t_hand / F * 1e3, 3), tensors_on_device=bool(on_device),
                note="hmsg_crop_resize_batch -> torch module -> hmsg_add_frame_features by data_ptr(): no host copy of masks or features")
