"""The two distance forms of faiss (include/hmsg.h: hmsg_config.overlap_distance_form) on the configs[1] scene itself, on the MI355X:
the 1000-frame build with HMSG_OVERLAP_DIRECT and with HMSG_OVERLAP_FAISS_BLAS, every pair's (ratio, threshold) logged by the merge
(HMSG_DEBUG_OVERLAP_LOG).  Prints how many merge decisions differ (compared pair by pair while the two folds evaluate the same
pairs -- after a first flip the instance lists differ and so do the pairs), a histogram of |ratio - threshold| (the margin a
rounding would have to cross; for a pair the first direction decided, `ratio` is that direction's, a lower bound), and whether
the final instances are the same.
    gpurun -- 'python scripts/fuzz/faiss_form_configs1.py [frames=1000] > gpurun_out/faiss_form_configs1.txt'"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)


def run(form, frames, log):
    """one build in a process of its own (the log switch is read once per process)"""
    code = r"""
import sys, hashlib, numpy as np, torch
sys.path.insert(0, %r)
import bench
from holoagent_amd._lib import HmsgLib, Scene
from holoagent_amd.synth import SceneSpec
L = HmsgLib()
spec = SceneSpec(seed=1234, n_frames=%d, feat_dim=64, n_masks=32)
inp = bench.build_scene_inputs(L, spec, torch.device("cuda", 0), torch)
sc = Scene(lib_=L, height=spec.height, width=spec.width, max_frames=spec.n_frames, max_masks=32, feat_dim=64, overlap_distance_form=%d)
sc.add_frames(inp["rgb"], inp["depth"], inp["pose"], inp["K"]); sc.finalize_map()
sc.add_frame_features(0, inp["masks"], inp["f_g"], inp["f_masked"], inp["f_crop"]); sc.fuse_frames(); sc.merge_instances()
inst = sc.instances()
print(len(inst), hashlib.sha1(b"".join(np.ascontiguousarray(c).tobytes() for c in inst)).hexdigest(), float(np.abs(sc.map_points()).max()))
""" % (ROOT, frames, form)
    if os.path.exists(log):
        os.remove(log)
    env = dict(os.environ, HMSG_DEBUG_OVERLAP_LOG=log, HMSG_FOLD_NOPIPE="1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout.split()
    return int(out[0]), out[1], float(out[2]), np.fromfile(log, np.float64).reshape(-1, 2)


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    n0, h0, far, a = run(0, frames, "/tmp/ovlog_direct.bin")
    n1, h1, _, b = run(1, frames, "/tmp/ovlog_blas.bin")
    print("configs[1] scene, %d frames (largest |coordinate| of the map: %.1f m)" % (frames, far))
    print("direct form    : %6d instances, %8d pair evaluations, digest %s" % (n0, len(a), h0[:16]))
    print("faiss BLAS form: %6d instances, %8d pair evaluations, digest %s" % (n1, len(b), h1[:16]))
    n = min(len(a), len(b))
    same_th = a[:n, 1] == b[:n, 1]
    first_div = int(np.argmin(same_th)) if not same_th.all() else n
    da, db = a[:first_div, 0] > a[:first_div, 1], b[:first_div, 0] > b[:first_div, 1]
    print("pairs compared one to one (until the folds part ways): %d; merge decisions that differ among them: %d" % (first_div, int((da != db).sum())))
    print("final instances identical: %s" % (h0 == h1))
    m = np.abs(a[:, 0] - a[:, 1])
    edges = [0, 1e-4, 1e-3, 3e-3, 1e-2, 3e-2, 0.1, 0.3, 1.01]
    hist, _ = np.histogram(m, edges)
    print("margin |ratio - threshold| of the direct form's %d decisions:" % len(m))
    for lo, hi, c in zip(edges[:-1], edges[1:], hist):
        print("  [%7.4f, %7.4f): %8d" % (lo, hi, c))
    print("smallest margin: %.6f" % float(m.min()))


if __name__ == "__main__":
    main()
