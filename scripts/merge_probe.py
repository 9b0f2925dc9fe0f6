"""Development aid: the bench scene (configs[1] shape) up to the merge, with the library's debug output on stderr.
    HMSG_DEBUG_MERGESTATS=1 HMSG_DEBUG_TIMING=1 python scripts/merge_probe.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from holoagent_amd._lib import HmsgLib, Scene
from holoagent_amd.synth import SceneSpec

F = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
L = HmsgLib()
dev = torch.device("cuda", 0)
spec = SceneSpec(seed=1234, n_frames=F, feat_dim=512, n_masks=32)
inp = bench.build_scene_inputs(L, spec, dev, torch)
sc = Scene(lib_=L, device_id=0, height=spec.height, width=spec.width, max_frames=F, max_masks=32, feat_dim=512)
for r in range(reps):
    sc.reset()
    sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
    sc.finalize_map()
    sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"])
    sc.fuse_frames()
    torch.cuda.synchronize()
    t = time.perf_counter()
    sc.merge_instances()
    dt = time.perf_counter() - t
    inst = sc.instances()
    import hashlib
    hsh = hashlib.sha1()
    for a in inst:
        hsh.update(np.ascontiguousarray(a).tobytes())
    print("merge %.1f ms  instances %d  points %d  sha1 %s" % (dt * 1e3, len(inst), sum(len(a) for a in inst), hsh.hexdigest()))
