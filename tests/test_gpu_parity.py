"""Parity tests proper: the gfx950 HIP library driven through the C ABI against the CPU oracle."""
import numpy as np
import pytest

from tests import golden_io as GI
from tests import parity_common as PC

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from holoagent_amd._lib import HmsgLib
    return HmsgLib()          # raises if the HIP library is missing -- no fallback


def test_map_and_fuse_golden_inputs(L):
    z = GI.load("build_hier")
    frames = GI.unpack_frames(z)
    cfg = GI.unpack_cfg(z)
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"]))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    # the reference-produced cloud of the fixture is the same set too
    np.testing.assert_allclose(ref_pts, z["ref_cloud"], rtol=0, atol=1e-12)
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols)
    sc.close()


def test_map_and_fuse_medium(L):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=21, rooms_x=2, rooms_z=1, room_size=(4.0, 2.6, 3.5), objects_per_room=5, width=160,
                     height=120, n_frames=70, n_masks=20, feat_dim=256)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=256)
    sc = PC.make_scene(L, frames, dict(feat_dim=256))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    assert ref_pts.shape[0] > 1000
    PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks=True)
    sc.close()


@pytest.mark.parametrize("name", ["build_seq", "build_hier", "build_ragged"])
def test_full_build_against_oracle_and_reference_golden(L, name):
    """A1..A7 on the inputs of the reference-generated fixtures: stage-wise BIT-LEVEL parity with the oracle, then the
    end result against what the reference's own create_feature_map produced (tests/golden).  build_ragged has a
    different number of masks in every frame (1 .. 70), as SAM's output does."""
    z = GI.load(name)
    frames = GI.unpack_frames(z)
    cfg = GI.unpack_cfg(z)
    sc = PC.make_scene(L, frames, dict(feat_dim=cfg["feat_dim"], merge_type=1 if cfg["merge_type"] == "hierarchical" else 0))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    assert np.array_equal(ref_pts, z["ref_cloud"])                       # the reference run's cloud, bit for bit
    ref_feats, _ = PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks=(name != "build_seq"))
    got, feats = PC.check_merge_pool(sc, frames, cfg, ref_pts, ref_feats)
    PC.check_against_reference_run(name, z, sc, got, feats)
    sc.close()


def test_full_path_d1024(L):
    """configs[2] shape (1024-d features, DINOv2 ViT-L width): A1..A7 stage-wise against the oracle on a small scene."""
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=33, rooms_x=1, rooms_z=1, room_size=(4.0, 2.6, 3.5), objects_per_room=5, width=160,
                     height=120, n_frames=24, n_masks=16, feat_dim=1024)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=1024,
               init_overlap_thresh=0.75, overlap_thresh_factor=0.025, iou_thresh=0.05, merge_type="sequential",
               outlier_nb=300)
    sc = PC.make_scene(L, frames, dict(feat_dim=1024, outlier_nb_points=300))
    S, ref_pts, ref_cols = PC.check_map(sc, frames, cfg)
    ref_feats, _ = PC.check_fuse(sc, frames, S, cfg, ref_pts, ref_cols, check_masks=False)
    got, feats = PC.check_merge_pool(sc, frames, cfg, ref_pts, ref_feats)
    assert len(got) >= 3 and feats.shape[1] == 1024
    sc.close()


def test_query_golden(L):
    PC.check_query_golden(L)


@pytest.mark.parametrize("D", [512, 1024])
def test_query_random_large(L, D):
    """1k queries x 5k nodes x D (CLIP ViT-B/32 width and configs[2]'s 1024) against the numpy restatement
    (oracle.query_object)."""
    from holoagent_amd._lib import NodeIndex
    from oracle import hmsg_oracle as O
    rng = np.random.Generator(np.random.PCG64(77))
    N, R, Q, k = 5000, 40, 1000, 5
    emb = rng.standard_normal((N, D)) * 0.05
    room = rng.integers(0, R, size=N).astype(np.int32)
    T = rng.standard_normal((Q, 2, D)).astype(np.float32) * 0.05
    lists = [sorted(rng.choice(R, size=int(rng.integers(1, 6)), replace=False).tolist()) for _ in range(Q)]
    ix = NodeIndex(emb, room, lib_=L)
    idx, rooms, score = ix.query_objects(T, np.zeros(Q, np.int32), lists, k)
    for q in range(0, Q, 7):
        cand = [o for r in lists[q] for o in np.nonzero(room == r)[0]]
        top, sc = O.query_object(T[q], 0, emb[cand], k)
        assert [cand[t] for t in top] == [int(v) for v in idx[q] if v >= 0]
        np.testing.assert_allclose(score[q][: len(top)], sc, rtol=0, atol=1e-12)
    ix.close()


def test_graph_mirror_end_to_end(L, tmp_path):
    """holoagent_amd.graph.Graph (mirror of the reference's Graph): build -> assemble -> save (reference layout)
    -> load -> hierarchical query, on the GPU library."""
    from tests.test_emu_parity import test_graph_end_to_end_tiny
    test_graph_end_to_end_tiny.__wrapped__(L, tmp_path) if hasattr(test_graph_end_to_end_tiny, "__wrapped__") \
        else test_graph_end_to_end_tiny(L, tmp_path)


def test_merge_shortcuts_are_exact(L):
    """The merge fold's accelerations (anchor DBSCAN, fixed-point skipping) must not
    change a single point: the same scene merged with them disabled / forced gives bit-identical instances."""
    import os
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=5, rooms_x=2, rooms_z=1, room_size=(4.0, 2.6, 3.5), objects_per_room=5, width=320,
                     height=240, n_frames=120, n_masks=24, feat_dim=32)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    S = PC.stack_frames(frames)
    results = []
    for env in ({}, {"HMSG_DEBUG_NOANCHOR": "1"}):
        for k in ("HMSG_DEBUG_NOANCHOR",):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            sc = PC.make_scene(L, frames, dict(feat_dim=32))
            sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
            sc.finalize_map()
            sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"])
            sc.fuse_frames()
            sc.merge_instances()
            results.append([np.array(p) for p in sc.instances()])
            sc.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    base = results[0]
    assert len(base) > 10 and max(len(p) for p in base) > 4096
    for other in results[1:]:
        assert len(other) == len(base)
        for a, b in zip(base, other):
            assert a.shape == b.shape and np.array_equal(a, b)


_ACCUM_GPU_SNIPPET = """
import sys, json, hashlib
sys.path.insert(0, {root!r})
import numpy as np
from tests import parity_common as PC
from holoagent_amd._lib import HmsgLib
from holoagent_amd.synth import SceneSpec, SynthScene
L = HmsgLib()
spec = SceneSpec(seed=21, rooms_x=2, rooms_z=1, room_size=(4.0, 2.6, 3.5), objects_per_room=5, width=160,
                 height=120, n_frames=70, n_masks=4, feat_dim=16)
scn = SynthScene(spec)
frames = [scn.frame(i) for i in range(spec.n_frames)]
cfg = dict(voxel_size=0.05, clip_masked_weight=0.4418, max_mask_distance=10000, feat_dim=16)
sc = PC.make_scene(L, frames, dict(feat_dim=16))
PC.check_map(sc, frames, cfg)               # (bit-identical with the oracle's ordered float64 sums)
pts, cols = sc.map_points(colors=True)
print("DIGEST", json.dumps([int(pts.shape[0]), hashlib.sha1(pts.tobytes()).hexdigest()]))
"""


def test_ordered_voxel_sums_two_routes_gpu():
    """Round 6: the ordered float64 voxel sums with a pack's points staged in LDS and three lanes adding a coordinate each (the
    default), and with every lane carrying the sums (until round 5) -- each against the oracle, one cloud."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for over in ({}, {"HMSG_DEBUG_ACCUM_WAVES": "1"}):
        env = dict(os.environ)
        env.pop("HMSG_DEBUG_ACCUM_WAVES", None)
        env.update(over)
        r = subprocess.run([sys.executable, "-c", _ACCUM_GPU_SNIPPET.format(root=root)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout + r.stderr
        out.append(json.loads([l for l in r.stdout.splitlines() if l.startswith("DIGEST")][-1][7:]))
    assert out[0] == out[1] and out[0][0] > 1000, out
