"""The encoder hand-off (sam_clip_feats_extractor.py:117-158, clip_utils.py:63-94): features made by a live PyTorch-ROCm
module reach hmsg_add_frame_features by DEVICE pointer (`tensor.data_ptr()`), masks likewise, and give bit for bit the map
features that the same numbers handed over from host memory give."""
import numpy as np
import pytest

from tests import parity_common as PC


@pytest.mark.gpu
def test_features_by_device_pointer_equal_features_from_the_host():
    import torch
    from holoagent_amd._lib import HmsgLib
    from holoagent_amd.encoder_handoff import make_vit_b32
    from holoagent_amd.synth import SceneSpec, SynthScene
    L = HmsgLib()
    dev = torch.device("cuda", 0)
    spec = SceneSpec(seed=21, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=128, height=96, n_frames=6, n_masks=8,
                     feat_dim=64)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    S = PC.stack_frames(frames)
    torch.manual_seed(1)
    enc = make_vit_b32(torch, dim_out=64, width=128, layers=2, heads=4, patch=32, image=64).to(dev).half().eval()
    masks_d = torch.from_numpy(S["masks"]).to(dev)                      # [F, M, H, W] u8, resident
    rgb_d = torch.from_numpy(S["rgb"]).to(dev)
    feats = []
    with torch.no_grad():
        for f in range(spec.n_frames):
            img = torch.nn.functional.interpolate(rgb_d[f].permute(2, 0, 1)[None].float(), size=(64, 64))      # stand-in for the crops
            x = img.repeat(2 * 8 + 1, 1, 1, 1) * (1.0 + 0.01 * torch.arange(17, device=dev).view(-1, 1, 1, 1)) / 255.0
            e = torch.nn.functional.normalize(enc(x.half()).float(), dim=-1).contiguous()
            feats.append((e[:1].contiguous(), e[1:9].contiguous(), e[9:].contiguous()))
    out = []
    for on_device in (True, False):
        sc = PC.make_scene(L, frames, dict(feat_dim=64, outlier_nb_points=200))
        sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
        sc.finalize_map()
        for f, (fg, fm, fc) in enumerate(feats):
            if on_device:
                assert fg.is_cuda and fm.is_cuda and fc.is_cuda and masks_d.is_cuda
                sc.add_frame_features(f, masks_d[f][None], fg, fm[None], fc[None])
            else:
                sc.add_frame_features(f, S["masks"][f][None], fg.cpu().numpy(), fm.cpu().numpy()[None], fc.cpu().numpy()[None])
        sc.fuse_frames()
        out.append(sc.map_feats().copy())
        sc.close()
    assert out[0].shape[0] > 100 and np.abs(out[0]).sum() > 0
    assert np.array_equal(out[0], out[1])
