"""hmsg_kmeans (include/hmsg.h; holoagent_amd/csrc/hmsg_kmeans.hip) against scikit-learn itself -- the reference's own call,
utils/graph_utils.py:329-333: KMeans(n_clusters=num_views, max_iter=100, n_init=5, random_state=0).fit(room_clip_embeddings).
scikit-learn is the oracle: labels must be equal, centres within float32 rounding (they come out bit-equal on this image)."""
import os
import warnings

import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC


def _lib():
    from holoagent_amd._lib import HmsgLib
    return HmsgLib(PC.EMU_PATH if os.path.exists(PC.EMU_PATH) else None)     # (host code: either build carries it)


def _sklearn(X, k):
    from sklearn.cluster import KMeans
    from threadpoolctl import threadpool_limits
    with threadpool_limits(limits=1), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return KMeans(n_clusters=k, max_iter=100, n_init=5, random_state=0).fit(X)


def _cases():
    rng = np.random.default_rng(7)
    for trial in range(9):
        n, D, k = int(rng.integers(24, 260)), int(rng.choice([16, 64, 512])), int(rng.choice([5, 24]))
        if trial % 3 == 0:
            X = rng.standard_normal((n, D))
        elif trial % 3 == 1:                                    # unit rows around a few directions (what CLIP features look like)
            c = rng.standard_normal((7, D))
            X = c[rng.integers(0, 7, n)] + 0.3 * rng.standard_normal((n, D))
            X /= np.linalg.norm(X, axis=1, keepdims=True)
        else:                                                   # repeated rows: fewer distinct points than clusters is possible
            X = rng.standard_normal((n, D))
            X[::3] = X[0]
        yield np.ascontiguousarray(X, np.float32), k


def test_kmeans_equals_scikit_learn_on_random_data():
    from holoagent_amd._lib import kmeans
    L = _lib()
    for X, k in _cases():
        km = _sklearn(X, k)
        labels, centers, inertia, n_iter = kmeans(X, k, lib_=L)
        assert np.array_equal(labels, km.labels_), (X.shape, k)
        np.testing.assert_allclose(centers, km.cluster_centers_, rtol=0, atol=1e-6)
        assert n_iter == km.n_iter_ and abs(inertia - km.inertia_) <= 1e-5 * max(1.0, abs(km.inertia_))


def test_kmeans_on_the_room_fixture_rows():
    """the rows compute_room_embeddings clusters in the reference-made fixture (tests/golden/roomemb.npz: a run of the
    reference's own compute_room_embeddings with num_views = 24): hmsg_kmeans + hmsg_pick_representative_views give the
    representative images the reference run picked."""
    from holoagent_amd._lib import kmeans, pick_representative_views
    L = _lib()
    z = GI.load("roomemb")
    done = 0
    for r in range(int(z["n_rooms"])):
        ids, X = np.asarray(z["img_ids_%d" % r], np.int64), np.ascontiguousarray(z["clip_%d" % r], np.float32)
        if len(ids) < 24:
            continue
        km = _sklearn(X, 24)
        labels, centers, _, _ = kmeans(X, 24, lib_=L)
        assert np.array_equal(labels, km.labels_)
        np.testing.assert_allclose(centers, km.cluster_centers_, rtol=0, atol=1e-6)
        picked = pick_representative_views(X, labels, centers, lib_=L)
        got, want = [int(ids[p]) for p in picked], [int(v) for v in z["repr_ids_%d" % r]]
        # (a two-member cluster is an exact tie between its members, which the reference decides inside BLAS: include/hmsg.h)
        sizes = np.bincount(labels, minlength=24)
        present = [lab for lab in range(24) if sizes[lab]]
        assert len(got) == len(want)
        assert all(g == w for g, w, lab in zip(got, want, present) if sizes[lab] != 2)
        done += 1
    assert done >= 1


def test_kmeans_rejects_bad_arguments():
    from holoagent_amd._lib import HmsgError, kmeans
    L = _lib()
    with pytest.raises(HmsgError):
        kmeans(np.zeros((3, 4), np.float32), 5, lib_=L)        # fewer rows than clusters (scikit-learn raises too)


def test_kmeans_reproduces_the_reference_runs_representative_views():
    """tests/golden/rooms_frames.npz (round 5: 48 DISTINCT global features per room, so that KMeans(24) of the reference's
    compute_room_embeddings clusters real points -- round 4's fixture clustered a dozen repeated rows): hmsg_kmeans +
    hmsg_pick_representative_views on a room's sample images give the representative images the REFERENCE RUN picked with
    scikit-learn, cluster by cluster; only a two-member cluster may differ (an exact tie that the reference decides inside BLAS)."""
    from holoagent_amd._lib import kmeans, pick_representative_views
    L = _lib()
    z = np.load(os.path.join(GI.GOLDEN, "rooms_frames.npz"))
    f_g = np.asarray(z["f_g"], np.float32)
    assert len(np.unique(f_g, axis=0)) == len(f_g)                       # every frame has a feature of its own
    for r in range(int(z["n_rooms"])):
        ids, ref = np.asarray(z["sample_%d" % r], np.int64), [int(v) for v in z["represent_%d" % r]]
        assert len(ids) >= 24 and len(ref) == 24
        X = np.ascontiguousarray(f_g[ids])
        labels, centers, _, _ = kmeans(X, 24, lib_=L)
        sizes = np.bincount(labels, minlength=24)
        assert (sizes > 0).all()                                         # 24 distinct clusters: no duplicate-point warning in the run
        got = [int(ids[p]) for p in pick_representative_views(X, labels, centers, lib_=L)]
        assert len(got) == 24
        for lab, (g, w) in enumerate(zip(got, ref)):
            assert g == w or sizes[lab] == 2, (r, lab, g, w, int(sizes[lab]))
        assert sum(g == w for g, w in zip(got, ref)) >= 24 - int((sizes == 2).sum())
