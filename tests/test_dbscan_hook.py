"""The segmented keep-largest DBSCAN (hmsg_test_dbscan hook) against the oracle's pcd_denoise_dbscan on seeded random
clouds: uniform boxes, planar patches with exact duplicates, blobs; several clouds per batch; and the anchor hint of the
merge fold (a fixed single-cluster cloud with its persisted core flags + new points): kept points, core flags and the
bookkeeping triple (changed, clusters, contested) must not depend on the hint.  Simulator here, same code on the GPU."""
import os

import numpy as np
import pytest

from oracle import hmsg_oracle as O
from tests import parity_common as PC


def run(L, clouds, eps, mp, core0=None):
    K = len(clouds)
    sizes = np.array([len(c) for c in clouds], np.int64)
    N = int(sizes.sum())
    pts = np.ascontiguousarray(np.concatenate(clouds) if N else np.zeros((0, 3)))
    outp, outs = np.zeros((max(N, 1), 3)), np.zeros(K, np.int64)
    outc, info = np.zeros(max(N, 1), np.uint8), np.zeros((K, 3), np.int32)
    c0 = None if core0 is None else np.ascontiguousarray(core0, np.uint8)
    rc = L.c.hmsg_test_dbscan(pts.ctypes.data, K, sizes.ctypes.data, eps, mp, None if c0 is None else c0.ctypes.data,
                              outp.ctypes.data, outs.ctypes.data, outc.ctypes.data, info.ctypes.data)
    assert rc == 0
    off = np.concatenate([[0], np.cumsum(outs)])
    return [outp[off[k]:off[k + 1]] for k in range(K)], [outc[off[k]:off[k + 1]] for k in range(K)], info


def cloud(rng, n, kind):
    if kind == 0:
        return rng.uniform(0, rng.uniform(0.3, 2.0), (n, 3))
    if kind == 1:                                          # a re-observed surface: half the points on exact lattice sites
        p = np.zeros((n, 3))
        p[:, 0], p[:, 1], p[:, 2] = rng.uniform(0, 1.5, n), rng.uniform(0, 1.0, n), rng.normal(0, 0.004, n)
        m = rng.random(n) < 0.5
        p[m] = (np.round(p / 0.05) * 0.05)[m]
        return p
    c = rng.uniform(0, 2, (int(rng.integers(1, 5)), 3))
    return c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.02, 0.15), (n, 3))


def check(L, seed, rounds):
    rng = np.random.default_rng(seed)
    anchored = 0
    for _ in range(rounds):
        eps, mp = float(rng.choice([0.1, 0.05, 0.08])), int(rng.choice([10, 5, 3]))
        clouds = [cloud(rng, int(rng.integers(0, 1200)), int(rng.integers(0, 3))) for _ in range(int(rng.integers(1, 5)))]
        got, _, _ = run(L, clouds, eps, mp)
        for k, c in enumerate(clouds):
            assert np.array_equal(got[k], O.pcd_denoise_dbscan(c, None, eps, mp)[0]), (seed, k)
        A = got[int(np.argmax([len(g) for g in got]))]
        if len(A) < 50:
            continue
        _, c2, i2 = run(L, [A], eps, mp)
        if i2[0, 0] != 0 or i2[0, 1] != 1:                 # not a fixed single-cluster cloud: no anchor
            continue
        nb = int(rng.integers(1, 400))
        B = A[rng.integers(0, len(A), nb)] + rng.normal(0, rng.uniform(0.01, 0.2), (nb, 3))
        cat = np.concatenate([A, B])
        hint = np.concatenate([c2[0], np.zeros(len(B), np.uint8)])
        g3, c3, i3 = run(L, [cat], eps, mp, core0=hint)
        g4, c4, i4 = run(L, [cat], eps, mp)
        assert np.array_equal(g3[0], O.pcd_denoise_dbscan(cat, None, eps, mp)[0]) and np.array_equal(g4[0], g3[0])
        assert np.array_equal(c3[0], c4[0]) and (i3 == i4).all()
        anchored += 1
    return anchored


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_dbscan_random_clouds_simulator():
    from holoagent_amd._lib import HmsgLib
    assert check(HmsgLib(PC.EMU_PATH), 5, 6) >= 2


@pytest.mark.gpu
def test_dbscan_random_clouds_gpu():
    from holoagent_amd._lib import HmsgLib
    assert check(HmsgLib(), 6, 40) >= 15
