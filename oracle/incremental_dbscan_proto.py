"""TEST INFRASTRUCTURE / DESIGN PROTOTYPE (round 3 groundwork) -- not used by the product.

The merge fold's step is pcd_denoise_dbscan(A ++ B) (graph_utils.py:667-679, 827-880) where A is an ANCHOR: a cloud that is
a fixed point of that very DBSCAN with one cluster (every point kept last time; its exact core flags are known) and B are
the few new points of a frame's masks.  The batch kernels re-cluster all of A every step.  This module states the
INCREMENTAL step -- only neighbourhoods of B are looked at -- and returns everything the fold's bookkeeping needs;
tests/test_incremental_proto.py checks it against the batch DBSCAN (the library's own, through hmsg_test_dbscan, and the
oracle) on random inputs.  Spatial queries go through scipy's cKDTree here; on the device they become lookups in a
persistent per-anchor cell grid.

Precondition on A ("anchor"): EVERY point of A belongs to its one cluster.  A cloud that the batch pass returned
unchanged with one cluster satisfies it whenever min_points >= 5 (the reference uses 10): a cluster holds a core point
and its >= min_points neighbours, so the "largest cluster has fewer than 5 points -> return the input" branch of
pcd_denoise_dbscan (graph_utils.py:866-871) cannot be what left the cloud unchanged.

Why it is exact (Open3D ClusterDBSCAN semantics as restated in oracle/hmsg_oracle.py o3d_cluster_dbscan):
  * core status is monotone: adding points only raises neighbour counts, so A's cores stay core; an A point that was
    not core can be promoted only if it has a B point within eps -- its count is re-taken (old A neighbours + B);
  * A is one cluster, so all of A's old cores are connected; a promoted A point was a border point of that cluster, hence
    within eps of an old core, hence in the same component C_A; a B core is in C_A iff its component of the B-core graph
    touches (within eps) any core of A (old or promoted);
  * cluster ids follow the smallest core index and A comes first in the concatenation, so C_A has id 0; a border point
    takes the smallest id among the clusters whose cores reach it, so every non-core point with a C_A core within eps
    goes to C_A -- in particular every non-core point of A stays in C_A: ALL of A is kept;
  * the kept cluster is the largest one; clusters other than C_A consist of B points only, so C_A (>= |A| points) wins
    whenever |A| > |B| (otherwise the caller falls back to the batch path);
  * result = A (unchanged, same order) ++ [b in B, in order, labelled C_A].
"""
from __future__ import annotations

import numpy as np
from scipy.sparse import coo_matrix
from scipy.sparse.csgraph import connected_components
from scipy.spatial import cKDTree


def _d2(p, q):
    d = p - q
    return (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def _within(tree_pts, tree, q, eps):
    """indices of tree points strictly closer than eps to q (nanoflann radius search is strict, on squared distances)"""
    cand = np.asarray(tree.query_ball_point(q, eps * (1 + 1e-9) + 1e-12), dtype=np.int64)
    if cand.size == 0:
        return cand
    return cand[_d2(tree_pts[cand], q) < eps * eps]


def incremental_step(A, A_core, B, eps, min_points):
    """pcd_denoise_dbscan(A ++ B) for an anchor A.  Returns None when the preconditions do not hold (|A| <= |B|), else a
    dict: keep_B (bool per B point), core_A (updated flags), core_B (flags of all B points), changed, n_clusters,
    contested -- the same quantities the batch pass reports."""
    nA, nB = len(A), len(B)
    if nA <= nB or nA == 0:
        return None
    tA, tB = cKDTree(A), cKDTree(B) if nB else None
    nbrA_of_B = [_within(A, tA, B[j], eps) for j in range(nB)]                 # A points near each b
    nbrB_of_B = [_within(B, tB, B[j], eps) for j in range(nB)]                 # B points near each b (itself included)
    core_B = np.array([len(nbrA_of_B[j]) + len(nbrB_of_B[j]) >= min_points for j in range(nB)], bool)
    # A points with a B point within eps: the only ones whose status can change
    touched = np.unique(np.concatenate(nbrA_of_B)) if nB and any(len(x) for x in nbrA_of_B) else np.zeros(0, np.int64)
    core_A = np.asarray(A_core, bool).copy()
    for a in touched:
        if core_A[a]:
            continue
        cnt = len(_within(A, tA, A[a], eps)) + len(_within(B, tB, A[a], eps))
        core_A[a] = cnt >= min_points
    # components of the B-core graph, and which of them touch a core of A
    bc = np.nonzero(core_B)[0]
    pos = -np.ones(nB, np.int64)
    pos[bc] = np.arange(len(bc))
    rows, cols = [], []
    for j in bc:
        for k in nbrB_of_B[j]:
            if core_B[k] and k > j:
                rows.append(pos[j])
                cols.append(pos[k])
    ncomp, comp = (connected_components(coo_matrix((np.ones(len(rows), np.int8), (rows, cols)), shape=(len(bc), len(bc))),
                                        directed=False) if len(bc) else (0, np.zeros(0, np.int64)))
    comp_in_A = np.zeros(ncomp, bool)
    for j in bc:
        if core_A[nbrA_of_B[j]].any():
            comp_in_A[comp[pos[j]]] = True
    in_CA_core = np.zeros(nB, bool)                                            # B cores that belong to C_A
    in_CA_core[bc] = comp_in_A[comp[pos[bc]]]
    # non-core B points: kept iff a C_A core (of A or of B) is within eps
    keep_B = in_CA_core.copy()
    other_core_adjacent = np.zeros(nB, bool)                                   # ... and do cores of OTHER clusters reach it
    for j in np.nonzero(~core_B)[0]:
        reach_A = core_A[nbrA_of_B[j]].any() or in_CA_core[nbrB_of_B[j]].any()
        keep_B[j] = reach_A
        other_core_adjacent[j] = (core_B[nbrB_of_B[j]] & ~in_CA_core[nbrB_of_B[j]]).any()
    n_other = int((~comp_in_A).sum())
    n_clusters = 1 + n_other
    # contested (the batch pass: a border point whose adjacent cores belong to more than one cluster; only looked for when
    # there is more than one cluster): B border points between C_A and another cluster, B border points between two other
    # clusters, and A's non-core points that a core of another cluster reaches
    contested = False
    if n_other:
        for j in np.nonzero(~core_B)[0]:
            adj = nbrB_of_B[j][core_B[nbrB_of_B[j]]]
            ids = set(np.where(in_CA_core[adj], -1, comp[pos[adj]]).tolist())
            if core_A[nbrA_of_B[j]].any():
                ids.add(-1)
            if len(ids) > 1:
                contested = True
                break
        if not contested:
            for a in touched:
                if core_A[a]:
                    continue
                nb = _within(B, tB, A[a], eps)
                if (core_B[nb] & ~in_CA_core[nb]).any():                       # (it always has a C_A core within eps)
                    contested = True
                    break
    return dict(keep_B=keep_B, core_A=core_A, core_B=core_B, changed=bool((~keep_B).any()), n_clusters=n_clusters,
                contested=contested)
