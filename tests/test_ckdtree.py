"""The host restatement of scipy's cKDTree (holoagent_amd/csrc/hmsg_ckdtree.h) against scipy itself: same index
permutation after the default build, same node count, and the same answer to query(x, k=1) -- in particular on
BIT-EQUAL distance ties, where the answer depends on the traversal order."""
import ctypes as C
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

from tests import parity_common as PC

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


def run(L, pts, q):
    pts = np.ascontiguousarray(pts, np.float64)
    q = np.ascontiguousarray(q, np.float64)
    out = np.empty(len(q), np.int64)
    perm = np.empty(len(pts), np.int64)
    nn = C.c_int64(0)
    rc = L.c.hmsg_test_ckdtree(pts.ctypes.data, len(pts), q.ctypes.data, len(q), out.ctypes.data, perm.ctypes.data, C.byref(nn))
    assert rc == 0
    return out, perm, nn.value


def count_nodes(node):
    return 1 if node.split_dim == -1 else 1 + count_nodes(node.lesser) + count_nodes(node.greater)


@pytest.fixture(scope="module")
def L():
    from holoagent_amd._lib import HmsgLib
    return HmsgLib(PC.EMU_PATH)


def test_random_cloud(L):
    rng = np.random.Generator(np.random.PCG64(1))
    pts = rng.uniform(-3, 5, (20000, 3))
    q = rng.uniform(-4, 6, (5000, 3))
    tree = cKDTree(pts)
    out, perm, nn = run(L, pts, q)
    assert np.array_equal(perm, tree.indices)
    assert nn == count_nodes(tree.tree)
    assert np.array_equal(out, tree.query(q, k=1)[1])


def test_voxel_grid_with_exact_ties(L):
    """Voxel-centroid-like cloud (jittered 5 cm lattice with holes and duplicate coordinates) queried at exact midpoints
    of neighbouring points, at the points themselves and at lattice-symmetric positions: thousands of bit-equal ties."""
    rng = np.random.Generator(np.random.PCG64(2))
    g = np.stack(np.meshgrid(np.arange(40), np.arange(12), np.arange(30), indexing="ij"), -1).reshape(-1, 3)
    g = g[rng.random(len(g)) < 0.6]
    pts = g * 0.05 + np.where(rng.random((len(g), 1)) < 0.5, 0.0, rng.uniform(-0.01, 0.01, (len(g), 3)))
    tree = cKDTree(pts)
    i = rng.integers(0, len(pts), 6000)
    _, nb = tree.query(pts[i], k=2)
    mid = (pts[i] + pts[nb[:, 1]]) / 2            # exact midpoints of nearest pairs
    quad = pts[i] + np.array([0.025, 0.025, 0.0])  # equidistant from up to four lattice points
    q = np.concatenate([mid, quad, pts[i], g[rng.integers(0, len(g), 2000)] * 0.05 + 0.025])
    out, perm, nn = run(L, pts, q)
    assert np.array_equal(perm, tree.indices)
    assert nn == count_nodes(tree.tree)
    ref = tree.query(q, k=1)[1]
    d_ref = np.linalg.norm(pts[ref] - q, axis=1)
    # how many of these queries really are ties?
    d2, i2 = tree.query(q, k=2)
    assert int((d2[:, 0] == d2[:, 1]).sum()) > 1000
    assert np.array_equal(out, ref), int((out != ref).sum())


def test_fixture_cloud_pool_queries(L):
    """The reference run's own map cloud and instance clouds (tests/golden/build_seq): every pooling query
    (graph.py:456-458) answered like scipy answers it."""
    from oracle import hmsg_oracle as O
    from tests import golden_io as GI
    z = GI.load("build_seq")
    pts = z["ref_cloud"]
    tree = cKDTree(pts)
    off = z["ref_mask_off"]
    qs = [O.o3d_voxel_down_sample(z["ref_mask_pts"][off[k]:off[k + 1]], None, 0.05)[0] for k in range(len(off) - 1)]
    q = np.concatenate(qs)
    out, perm, _ = run(L, pts, q)
    assert np.array_equal(perm, tree.indices)
    assert np.array_equal(out, tree.query(q, k=1)[1])
