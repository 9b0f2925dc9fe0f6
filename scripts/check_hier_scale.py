"""GPU check (development aid): hierarchical merge of a few hundred full-resolution frames, with and without the
anchor hint -- instances must be bit-identical, and the run must finish in seconds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from holoagent_amd._lib import HmsgLib, Scene
from holoagent_amd.synth import SceneSpec
import bench

F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
L = HmsgLib()
dev = torch.device("cuda", 0)
spec = SceneSpec(seed=4321, n_frames=F, feat_dim=64, n_masks=32)
inp = bench.build_scene_inputs(L, spec, dev, torch)
res = []
for env in ({}, {"HMSG_DEBUG_NOANCHOR": "1"}):
    os.environ.pop("HMSG_DEBUG_NOANCHOR", None)
    os.environ.update(env)
    sc = Scene(lib_=L, device_id=0, height=spec.height, width=spec.width, max_frames=F, max_masks=32, feat_dim=64,
               merge_type=1)
    sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
    sc.finalize_map()
    sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"])
    sc.fuse_frames()
    t = time.time()
    sc.merge_instances()
    dt = time.time() - t
    inst = [np.array(p) for p in sc.instances()]
    print("hierarchical merge of %d frames: %.2f s, %d instances, %d points (%s)" %
          (F, dt, len(inst), sum(len(p) for p in inst), "no anchors" if env else "anchors"))
    res.append(inst)
    sc.close()
assert len(res[0]) == len(res[1])
for a, b in zip(*res):
    assert a.shape == b.shape and np.array_equal(a, b)
print("identical")
