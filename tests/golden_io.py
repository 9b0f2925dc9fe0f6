"""Loader for the committed golden fixtures (tests/golden/*.npz, made by oracle/refdrive/gen_golden.py)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def unpack_frames(z):
    F, H, W = z["depth"].shape
    M = z["f_masked"].shape[1]
    masks = np.unpackbits(z["masks"], axis=-1)[..., :W].astype(bool)
    nm = z["n_masks"] if "n_masks" in z.files else np.full(F, M)
    frames = []
    for i in range(F):
        m = int(nm[i])
        frames.append(dict(rgb=z["rgb"][i], depth=z["depth"][i], pose=z["pose"][i], K=z["K"], masks=masks[i][:m],
                           f_g=z["f_g"][i], f_masked=z["f_masked"][i][:m], f_crop=z["f_crop"][i][:m]))
    return frames


def unpack_cfg(z):
    cfg = {}
    for k, v in zip(z["cfg_keys"], z["cfg_vals"]):
        v = str(v)
        try:
            cfg[str(k)] = int(v)
        except ValueError:
            try:
                cfg[str(k)] = float(v)
            except ValueError:
                cfg[str(k)] = v
    return cfg
