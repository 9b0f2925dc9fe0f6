"""A10's view <-> object topology on the device (hmsg_object_views) against check_object_in_view
(fsr_vln/memory/hmsg/utils/graph_utils.py:95-157) as the mirror restates it in numpy (holoagent_amd/graph.py; that host
path is what tests/test_objects_golden.py pins against the reference run's objects_views.json).  Cameras in front of,
beside, behind and far away from the objects, a camera whose image cuts objects in half, duplicate cameras (equal mean
depths: first wins) and an image of zero size.  The views are the frames the points were seen in, so points re-project
onto the image border to ~1e-14: the decisions only agree because the kernel evaluates the two matrix products as
numpy's BLAS does (fused multiply-add chains) -- with separate multiplies and adds 2 of 79 pairs count other points.  Kernel simulator here; the switch `pipeline.views_on_device` makes
Graph.segment_hmsg_objects use the device batch and must give the same View / Object lists as the host path."""
import os

import numpy as np
import pytest

from tests import parity_common as PC


def _scene(L, n_frames=6):
    from holoagent_amd.synth import SceneSpec, SynthScene
    spec = SceneSpec(seed=61, rooms_x=1, rooms_z=1, room_size=(3.4, 2.6, 3.0), objects_per_room=4, width=96, height=72,
                     n_frames=n_frames, n_masks=6, feat_dim=16, yaw_step_deg=50.0)
    scn = SynthScene(spec)
    frames = [scn.frame(i) for i in range(spec.n_frames)]
    sc = PC.make_scene(L, frames, dict(feat_dim=16, outlier_nb_points=40, outlier_radius=0.5, feat_dbscan_min=20))
    S = PC.stack_frames(frames)
    sc.add_frames(S["rgb"], S["depth"], S["pose"], S["K"])
    sc.finalize_map()
    sc.add_frame_features(0, S["masks"], S["f_g"], S["f_masked"], S["f_crop"])
    sc.fuse_frames()
    sc.merge_instances()
    sc.pool_instances()
    return sc, frames, S


def _cameras(frames, rng):
    poses = [np.asarray(f["pose"], np.float64) for f in frames]
    out = list(poses)
    back = poses[0].copy()
    back[:3, :3] = back[:3, :3] @ np.diag([-1.0, 1.0, -1.0])              # looking the other way
    far = poses[1].copy()
    far[:3, 3] -= far[:3, 2] * 14.0                                        # 14 m back along the optical axis: mean depth > 10
    side = poses[2].copy()
    side[:3, 3] += side[:3, 0] * 0.9                                       # shifted: objects cut by the image border
    out += [back, far, side, poses[0].copy(), poses[3] + 0.0]
    for _ in range(4):                                                     # random nearby poses
        T = poses[int(rng.integers(len(poses)))].copy()
        T[:3, 3] += rng.uniform(-0.8, 0.8, 3)
        out.append(T)
    return out


def check_device_equals_host(L):
    from holoagent_amd.graph import check_object_in_view
    sc, frames, S = _scene(L)
    rng = np.random.default_rng(5)
    inst = sc.instances()
    assert len(inst) >= 4
    cams = _cameras(frames, rng)
    K = np.asarray(S["K"], np.float64)
    wh = [[96, 72]] * len(cams)
    wh[2] = [40, 72]                                                       # a narrower image for one camera
    wh[-1] = [0, 0]                                                        # nothing is inside an empty image
    inv = np.stack([np.linalg.inv(T) for T in cams])
    pi, pv = np.meshgrid(np.arange(len(inst)), np.arange(len(cams)), indexing="ij")
    vis, md = sc.object_views(inv, wh, K, pi.ravel(), pv.ravel())
    ref = [check_object_in_view(wh[v][0], wh[v][1], K, inv[v], np.asarray(inst[i], np.float64)) for i, v in zip(pi.ravel(), pv.ravel())]
    rv = np.array([bool(r[0]) for r in ref])
    rd = np.array([float(r[1]) for r in ref])
    assert np.array_equal(vis, rv), np.nonzero(vis != rv)
    fin = np.isfinite(rd)
    assert np.array_equal(np.isfinite(md), fin)
    np.testing.assert_allclose(md[fin], rd[fin], rtol=1e-13, atol=0)
    # every outcome is exercised: visible, not enough of it inside, behind the camera (inf), too far (finite depth > 10)
    assert rv.any() and (~rv & ~fin).any() and (~rv & fin & (rd > 10.0)).any()
    # duplicate cameras give equal mean depths (the first of them is the best view)
    dup = len(frames) + 3
    assert np.array_equal(md.reshape(len(inst), -1)[:, 0], md.reshape(len(inst), -1)[:, dup])
    # out-of-range pairs are refused
    from holoagent_amd._lib import HmsgError
    with pytest.raises(HmsgError):
        sc.object_views(inv, wh, K, [len(inst)], [0])
    with pytest.raises(HmsgError):
        sc.object_views(inv, wh, K, [0], [len(cams)])
    v0, m0 = sc.object_views(inv, wh, K, [], [])
    assert len(v0) == 0 and len(m0) == 0
    sc.close()


def check_graph_switch(L):
    """Graph.segment_hmsg_objects with pipeline.views_on_device: same objects, view lists and best views as the host path."""
    from holoagent_amd.graph import Graph
    sc, frames, S = _scene(L)

    class DS:
        def get_camera_intrinsics(self):
            return np.asarray(S["K"], np.float64)

        def __getitem__(self, i):
            return np.asarray(frames[i]["rgb"]), None, np.asarray(frames[i]["pose"], np.float64), None, None
    P = sc.map_points()
    lo, hi = P[:, [0, 2]].min(axis=0), P[:, [0, 2]].max(axis=0)
    gx, gz = np.meshgrid(np.arange(lo[0], hi[0], 0.05), np.arange(lo[1], hi[1], 0.05))
    verts = np.column_stack([gx.ravel(), gz.ravel()])
    out = []
    for on in (False, True):
        g = Graph.from_scene(sc, cfg=dict(main=dict(), models=dict(clip=dict(feat_dim=16)), pipeline=dict(views_on_device=on)), lib=L)
        g.dataset = DS()
        g.segment_floors_manually(None)
        g.set_rooms([dict(floor=0, vertices=verts, view_frames=list(range(len(frames))))])
        g.segment_hmsg_objects()
        out.append(([(o.object_id, list(o.view_ids), o.best_view_id) for o in g.objects],
                    [(v.view_id, list(v.object_ids), list(v.text_discription)) for v in g.views]))
    assert out[0] == out[1]
    assert len(out[0][0]) >= 3 and sum(len(o[1]) for o in out[0][0]) >= 3
    sc.close()


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_object_views_device_equals_host_emu():
    from holoagent_amd._lib import HmsgLib
    check_device_equals_host(HmsgLib(PC.EMU_PATH))


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_graph_views_on_device_switch_emu():
    from holoagent_amd._lib import HmsgLib
    check_graph_switch(HmsgLib(PC.EMU_PATH))


@pytest.mark.gpu
def test_object_views_gpu():            # (first MI355X run: gpurun_out/r04a, round 4's first GPU call -- green)
    from holoagent_amd._lib import HmsgLib
    check_device_equals_host(HmsgLib())
    check_graph_switch(HmsgLib())
