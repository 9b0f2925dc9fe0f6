"""Round-3 groundwork: the INCREMENTAL merge-fold step (oracle/incremental_dbscan_proto.py: only the neighbourhoods of
the new points are looked at) against the batch keep-largest DBSCAN of the library (hmsg_test_dbscan on the simulator)
and of the oracle, on random anchors + new points: kept points, core flags, and the bookkeeping triple."""
import os

import numpy as np
import pytest

from oracle import hmsg_oracle as O
from oracle.incremental_dbscan_proto import incremental_step
from tests import parity_common as PC
from tests.test_dbscan_hook import cloud, run

pytestmark = pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")


def check(L, seed, rounds):
    rng = np.random.default_rng(seed)
    done = multi = dropped = 0
    while done < rounds:
        eps, mp = float(rng.choice([0.1, 0.05, 0.08])), int(rng.choice([10, 5, 3]))
        raw = cloud(rng, int(rng.integers(100, 1200)), int(rng.integers(0, 3)))
        A = run(L, [raw], eps, mp)[0][0]
        if len(A) < 50:
            continue
        _, cA, iA = run(L, [A], eps, mp)
        if iA[0, 0] != 0 or iA[0, 1] != 1:                 # not a fixed single-cluster cloud: not an anchor
            continue
        if (O.o3d_cluster_dbscan(A, eps, mp) != 0).any():   # min_points < 5 only: "largest cluster has fewer than 5 points,
            continue                                        # the input is returned" also reports unchanged / one cluster
        nb = int(rng.integers(1, min(len(A), 300)))
        B = A[rng.integers(0, len(A), nb)] + rng.normal(0, rng.uniform(0.005, 0.25), (nb, 3))
        if rng.random() < 0.4:                              # a separate blob: a second cluster, maybe touching
            k = int(rng.integers(3, min(60, len(A) - nb) if len(A) - nb > 3 else 4))
            B = np.concatenate([B, A[rng.integers(0, len(A))] + rng.uniform(0.05, 0.4) * np.array([1.0, 0, 0]) +
                                rng.normal(0, 0.02, (k, 3))])
        if len(B) >= len(A):
            continue
        cat = np.concatenate([A, B])
        got, gcore, info = run(L, [cat], eps, mp)
        assert np.array_equal(got[0], O.pcd_denoise_dbscan(cat, None, eps, mp)[0])
        r = incremental_step(A, cA[0], B, eps, mp)
        want = np.concatenate([A, B[r["keep_B"]]])
        assert np.array_equal(got[0], want), (seed, done)
        assert np.array_equal(gcore[0].astype(bool), np.concatenate([r["core_A"], r["core_B"][r["keep_B"]]])), (seed, done)
        assert (int(r["changed"]), r["n_clusters"]) == (int(info[0, 0]), int(info[0, 1])), (seed, done, r["n_clusters"], info[0])
        if info[0, 1] > 1:
            assert int(r["contested"]) == int(info[0, 2]), (seed, done)
            multi += 1
        dropped += int(r["changed"])
        done += 1
    return multi, dropped


def test_incremental_step_equals_batch_dbscan():
    from holoagent_amd._lib import HmsgLib
    multi, dropped = check(HmsgLib(PC.EMU_PATH), 3, 25)
    assert multi >= 2 and dropped >= 5
