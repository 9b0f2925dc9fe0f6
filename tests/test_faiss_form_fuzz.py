"""How often can faiss's BLAS distance form flip an overlap decision of the merge?  (SURVEY A6 hazard iv, CPU only.)

find_overlapping_ratio_faiss (graph_utils.py:620-662) asks faiss 1.7.2 `IndexFlatL2.search(k=1)` for exact squared
float32 distances and counts `D < radius**2`.  For 20 or more queries faiss does not evaluate (dx*dx + dy*dy) + dz*dz per
pair: it takes the BLAS route |x|^2 + |y|^2 - 2 x.y (sgemm) and clamps at zero -- with coordinates of ~10 m the norms
are ~100 and carry a float32 rounding of ~1e-5, against r^2 = 5.6e-3.  faiss is not in this image (nor vendored in the
reference): the oracle and the HIP path use the direct form.  This test RUNS the sequential merge of the three
reference-made fixtures with both forms side by side on every pair the merge evaluates and reports

  * per point: how many `D < r^2` decisions differ between the two forms,
  * per pair:  how many merge decisions (`ratio > threshold`) differ -- the only thing that could change an instance.

It asserts that no merge decision flips on the fixtures (and prints the margins); it cannot prove that for other scenes.
The BLAS form is evaluated in float32 with numpy's sgemm, whose summation order need not be MKL's / OpenBLAS's of the
reference's wheel: the counts are an estimate of the knife edge's width, not a replay.
"""
import numpy as np
import pytest

from oracle import hmsg_oracle as O
from tests import golden_io as GI


def _blas_form_nn_sqdist(q, b):
    q = np.ascontiguousarray(q, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    qn = (q * q).sum(axis=1, dtype=np.float32)
    bn = (b * b).sum(axis=1, dtype=np.float32)
    out = np.empty(len(q), np.float32)
    for s in range(0, len(q), 2048):                      # (faiss blocks the queries as well)
        ip = q[s:s + 2048] @ b.T                          # float32 sgemm
        d = (qn[s:s + 2048, None] + bn[None, :]) - np.float32(2.0) * ip
        np.maximum(d, np.float32(0.0), out=d)
        out[s:s + 2048] = d.min(axis=1)
    return out


@pytest.mark.parametrize("name", ["build_seq", "build_ragged", "build_hier"])
def test_blas_form_does_not_flip_a_merge_decision_on_the_fixtures(name):
    z = GI.load(name)
    frames = GI.unpack_frames(z)[:10]                     # (the brute-force BLAS form is quadratic: ten frames, ~a minute)
    cfg = GI.unpack_cfg(z)
    stat = run_both_forms(frames, cfg)
    print("%s: %d pairs, %d point decisions, %d differ between the direct and the BLAS form (%.2e); merge decisions that "
          "differ: %d; smallest |ratio - threshold| = %.4f" % (name, stat["pairs"], stat["points"], stat["point_flips"],
                                                              stat["point_flips"] / max(1, stat["points"]), stat["pair_flips"], stat["min_margin"]))
    assert stat["pairs"] > 30
    assert stat["pair_flips"] == 0


def run_both_forms(frames, cfg):
    """the sequential merge of `frames` with the direct and the BLAS distance form side by side on every pair it evaluates
    (scripts/fuzz/fuzz_faiss_form.py runs this over random scenes; the counts of its last run are in profiles/)"""
    res = O.create_feature_map(frames, dict(cfg, merge_type="sequential"), keep_intermediates=True)
    frames_pcd = res["frames_pcd"]
    stat = dict(pairs=0, points=0, point_flips=0, pair_flips=0, min_margin=np.inf)
    th = cfg["init_overlap_thresh"]
    orig = O.find_overlapping_ratio

    def both(p1, p2, radius):
        direct = orig(p1, p2, radius)
        if p1.shape[0] == 0 or p2.shape[0] == 0:
            return direct
        r2 = np.float32(radius ** 2)
        n = []
        for a, b in ((p1, p2), (p2, p1)):
            dd = O.faiss_flat_l2_nn_sqdist(a, b) < r2
            db = _blas_form_nn_sqdist(a, b) < r2 if len(a) >= 20 else dd       # (below 20 queries faiss uses the direct form)
            stat["points"] += len(a)
            stat["point_flips"] += int((dd != db).sum())
            n.append(db.sum() / len(a))
        blas = max(n)
        stat["pairs"] += 1
        stat["pair_flips"] += int((direct > th) != (blas > th))
        stat["min_margin"] = min(stat["min_margin"], abs(direct - th))
        return direct

    O.find_overlapping_ratio = both
    try:
        O.seq_merge(frames_pcd, th, cfg["voxel_size"], cfg["iou_thresh"])
    finally:
        O.find_overlapping_ratio = orig
    return stat
