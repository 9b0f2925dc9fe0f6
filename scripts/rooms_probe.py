"""Print the room segmentation of a device-rendered scene (N1) as text and time it: python scripts/rooms_probe.py [frames] [res]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from holoagent_amd._lib import HmsgLib, Scene
from holoagent_amd.synth import SceneSpec

F = int(sys.argv[1]) if len(sys.argv) > 1 else 288
res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.05
L = HmsgLib()
spec = SceneSpec(seed=1234, n_frames=F, feat_dim=64, n_masks=32)
inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
sc = Scene(lib_=L, height=spec.height, width=spec.width, max_frames=F, max_masks=spec.n_masks, feat_dim=spec.feat_dim)
sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
sc.finalize_map()
P = sc.map_points()
lo, hi = float(P[:, 1].min()), float(P[:, 1].max())
for _ in range(3):
    t = time.perf_counter()
    m, n, xz = sc.segment_rooms(lo, hi, lo, hi - lo, res)
    dt = time.perf_counter() - t
print("map %d points, grid %s, %d rooms, %.2f ms" % (len(P), m.shape, n, dt * 1e3))
try:
    from oracle import rooms_oracle as R
    t = time.perf_counter()
    mo, no, _ = R.segment_rooms(P, lo, hi - lo, res)
    print("oracle %.1f ms, equal %s" % ((time.perf_counter() - t) * 1e3, np.array_equal(m, mo)))
except ImportError:
    pass
sy, sx = max(1, m.shape[0] // 60), max(1, m.shape[1] // 150)
for r in range(0, m.shape[0], sy):
    print("".join("#" if m[r, c] == n + 1 else ("." if m[r, c] <= 0 else chr(ord("0") + m[r, c])) for c in range(0, m.shape[1], sx)))
