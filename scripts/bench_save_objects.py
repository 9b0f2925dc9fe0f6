"""Row N2 micro-benchmark: writing the object level of a built scene to disk -- the library's bulk writer
(hmsg_save_objects through Graph.save_hmsg_graph) against the per-node Python writer (Object.save) on a sample.
    python scripts/bench_save_objects.py [--frames 300] [--sample 200]          # one JSON line
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from holoagent_amd._lib import HmsgLib, Scene  # noqa: E402
from holoagent_amd.graph import Graph  # noqa: E402
from holoagent_amd.synth import SceneSpec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--sample", type=int, default=200)
    a = ap.parse_args()
    L = HmsgLib()
    device = torch.device("cuda", 0)
    spec = SceneSpec(seed=1234, n_frames=a.frames, feat_dim=512, n_masks=32)
    inp = bench.build_scene_inputs(L, spec, device, torch)
    sc = Scene(lib_=L, device_id=0, height=spec.height, width=spec.width, max_frames=a.frames, max_masks=32, feat_dim=512)
    sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"])
    sc.finalize_map()
    sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    rooms = []
    for lo6 in inp["rooms"]:
        xs, zs = np.arange(lo6[0], lo6[3], 0.05), np.arange(lo6[2], lo6[5], 0.05)
        rooms.append(dict(floor=0, vertices=np.stack(np.meshgrid(xs, zs, indexing="ij"), -1).reshape(-1, 2)))
    g = Graph.from_scene(sc, lib=L)
    g.build_hier_multimodal_scene_graph(None, rooms=rooms)
    out = tempfile.mkdtemp(prefix="hmsg_save_")
    try:
        t0 = time.perf_counter()
        g.save_hmsg_graph(os.path.join(out, "bulk"))
        t_bulk = time.perf_counter() - t0
        n_obj = len(g.objects)
        files = os.listdir(os.path.join(out, "bulk", "objects"))
        size = sum(os.path.getsize(os.path.join(out, "bulk", "objects", f)) for f in files)
        os.makedirs(os.path.join(out, "py"))
        sample = g.objects[:a.sample]
        t0 = time.perf_counter()
        for o in sample:
            o.save(os.path.join(out, "py"))
        t_py = time.perf_counter() - t0
        same = all(open(os.path.join(out, "py", f), "rb").read() == open(os.path.join(out, "bulk", "objects", f), "rb").read()
                   for f in os.listdir(os.path.join(out, "py")))
        print(json.dumps({"metric": "objects_saved_per_s", "value": n_obj / t_bulk, "unit": "objects/s", "objects": n_obj,
                          "bytes": size, "seconds_whole_graph": t_bulk, "MB_per_s": size / t_bulk / 1e6,
                          "python_writer": {"objects": len(sample), "seconds": t_py, "objects_per_s": len(sample) / t_py,
                                            "byte_identical": bool(same)},
                          "speedup": (n_obj / t_bulk) / (len(sample) / t_py)}))
    finally:
        shutil.rmtree(out, ignore_errors=True)
    sc.close()


if __name__ == "__main__":
    main()
