"""Stage timing probe on the GPU (development aid): config-2-shaped inputs, fewer frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from holoagent_amd.synth import SceneSpec, SynthScene
from holoagent_amd._lib import Scene
from tests import parity_common as PC

F = int(sys.argv[1]) if len(sys.argv) > 1 else 128
D = int(sys.argv[2]) if len(sys.argv) > 2 else 512
spec = SceneSpec(seed=1234, n_frames=F, feat_dim=D, n_masks=32)
scn = SynthScene(spec)
t = time.time()
frames = [scn.frame(i) for i in range(F)]
print("gen %.1fs" % (time.time() - t))
S = PC.stack_frames(frames)
sc = Scene(height=spec.height, width=spec.width, max_frames=F, max_masks=32, feat_dim=D)
def T(name, fn):
    t = time.time(); fn(); dt = time.time() - t
    print("%-22s %8.2f ms  (%.1f frames/s)" % (name, dt * 1e3, F / dt)); return dt
T("add_frames(H2D)", lambda: sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"]))
T("finalize_map", sc.finalize_map)
print("V0 =", sc.map_size_unfiltered(), "V =", sc.map_size())
T("add_frame_features", lambda: sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"]))
T("fuse_frames", sc.fuse_frames)
nn = sc.frame_nn(F // 2)
print("valid px", (nn >= 0).mean())
T("merge_instances", sc.merge_instances)
inst = sc.instances()
print("instances", len(inst), "points", sum(len(i) for i in inst), "max", max(len(i) for i in inst))
T("pool_instances", sc.pool_instances)
