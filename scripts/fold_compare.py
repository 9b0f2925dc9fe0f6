"""Development aid: incremental fold vs batch fold (HMSG_FOLD_LEGACY=1) on a synthetic scene, bit for bit.
    python scripts/fold_compare.py [emu|gpu] [frames] [width] [height] [masks] [seed]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from holoagent_amd.synth import SceneSpec, SynthScene
from holoagent_amd._lib import HmsgLib
from tests import parity_common as PC

mode = sys.argv[1] if len(sys.argv) > 1 else "emu"
F = int(sys.argv[2]) if len(sys.argv) > 2 else 16
W = int(sys.argv[3]) if len(sys.argv) > 3 else 128
H = int(sys.argv[4]) if len(sys.argv) > 4 else 96
M = int(sys.argv[5]) if len(sys.argv) > 5 else 8
seed = int(sys.argv[6]) if len(sys.argv) > 6 else 3
L = HmsgLib(PC.EMU_PATH) if mode == "emu" else HmsgLib()
spec = SceneSpec(seed=seed, rooms_x=1, rooms_z=1, room_size=(3.6, 2.5, 3.2), objects_per_room=4, width=W, height=H,
                 n_frames=F, n_masks=M, feat_dim=16)
scn = SynthScene(spec)
frames = [scn.frame(i) for i in range(F)]

def run(mode):
    for k in ("HMSG_FOLD_LEGACY", "HMSG_FOLD_INCREMENTAL", "HMSG_FOLD_SWITCH"):
        os.environ.pop(k, None)
    if mode == "legacy":
        os.environ["HMSG_FOLD_LEGACY"] = "1"
    elif mode == "incremental":
        os.environ["HMSG_FOLD_INCREMENTAL"] = "1"
    else:
        os.environ["HMSG_FOLD_SWITCH"] = str(mode)
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=200, feat_dbscan_min=20))
    S = PC.stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"])
    sc.fuse_frames()
    t = time.time()
    sc.merge_instances()
    dt = time.time() - t
    inst = sc.instances()
    sc.close()
    return inst, dt

a, ta = run("legacy")
ok = True
for mode in ("incremental", 6000):
    b, tb = run(mode)
    same = len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b))
    print("batch fold %d instances (%.2fs)   %s: %d instances (%.2fs)  %s" % (len(a), ta, "incremental fold" if mode == "incremental" else "switch at %d points" % mode,
                                                                          len(b), tb, "IDENTICAL" if same else "DIFFERENT"))
    if not same:
        print("sizes batch", [len(x) for x in a])
        print("sizes other", [len(x) for x in b])
    ok = ok and same
sys.exit(0 if ok else 1)
