"""Retrieval GEMM at the size VERDICT r01 names for configs[2]: Q*C = 2000 text rows x N = 8000 nodes x D = 1024 (and
D = 512), float64 MFMA (v_mfma_f64_16x16x4_f64), timed with HIP events on the index's own stream."""
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (one HIP runtime for the process)
from holoagent_amd._lib import HmsgLib, NodeIndex

L = HmsgLib()
rng = np.random.Generator(np.random.PCG64(1))
out = {}
for D in (512, 1024):
    N, Q, C, k = 8000, 1000, 2, 5
    emb = rng.standard_normal((N, D)) * 0.05
    room = rng.integers(0, 40, size=N).astype(np.int32)
    T = rng.standard_normal((Q, C, D)).astype(np.float32) * 0.05
    lists = [[int(r)] for r in rng.integers(0, 40, Q)]
    ix = NodeIndex(emb, room, lib_=L)
    ix.query_objects(T, np.zeros(Q, np.int32), lists, k)          # warm-up
    ix.set_profiling(True)
    for _ in range(5):
        ix.query_objects(T, np.zeros(Q, np.int32), lists, k)
    n, ms, fl = ix.profile()
    out["D%d" % D] = dict(launches=n, ms_per_launch=round(ms / n, 4), tflops=round(fl / (ms * 1e-3) / 1e12, 2),
                          frac_of_f64_mfma_peak=round(fl / (ms * 1e-3) / 1e12 / 78.6, 3), flop_per_launch=fl / n)
    ix.close()
print(json.dumps(out))
