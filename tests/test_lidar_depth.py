"""Row N3: LiDAR -> depth images (include/hmsg.h: hmsg_lidar_depth) against the reference's own generate_depth.py outputs
(tests/golden/lidar_depth.npz, made by oracle/refdrive/gen_golden_depth.py) and against the oracle on fresh clouds."""
import os

import numpy as np
import pytest

from oracle import lidar_depth_oracle as LO
from tests import parity_common as PC

GOLD = os.path.join(os.path.dirname(__file__), "golden", "lidar_depth.npz")


def _gold():
    z = np.load(GOLD)
    W, H = (int(v) for v in z["WH"])
    return z, W, H


# ---------------------------------------------------------------------------------------------- oracle (CPU)
@pytest.mark.parametrize("tag,scale", [("s1", 1), ("s2", 2)])
def test_oracle_equals_reference_run(tag, scale):
    z, W, H = _gold()
    depth, flags = LO.occ_depth(z[tag + "_points_image"], z[tag + "_points_camera"], W, H, 1000, scale)
    assert np.array_equal(depth, z[tag + "_depth"])
    assert np.array_equal(flags, z[tag + "_flags"])
    assert flags.sum() > 1000 and (~flags).sum() > 1000 and (depth > 0).sum() > 3000


def test_oracle_projection_matches_reference_within_rounding():
    z, W, H = _gold()
    pi, pc = LO.project_points(z["points"], z["R"], z["t"], z["K"], W, H)
    # (the golden's first six image points were moved onto pixel centres by the generator)
    assert pc.shape == z["s1_points_camera"].shape
    np.testing.assert_allclose(pc, z["s1_points_camera"], rtol=0, atol=2e-15)
    np.testing.assert_allclose(pi[:, 6:], z["s1_points_image"][:, 6:], rtol=0, atol=1e-12)


def test_oracle_speckle_and_dilate_semantics():
    img = np.zeros((40, 60), np.int16)
    img[2:12, 2:12] = 5                  # 100 pixels: a speckle
    img[0:40, 20:50] = 7                 # 1200 pixels: stays
    img[20:30, 50:58] = 8                # touches the big region with |diff| = 1: joins it
    img[30:36, 0:10] = 9                 # 60 pixels, isolated
    out = LO.filter_speckles(img.copy(), 0, 1000, 1)
    assert out[5, 5] == 0 and out[31, 3] == 0 and out[10, 30] == 7 and out[25, 55] == 8
    src = np.zeros((24, 24), np.float32)
    src[10, 10] = 3.0
    # dst(x) = max src(x + o), o in [-8, +4]: the source pixel spreads 4 up / left and 8 down / right
    d = LO.dilate_rect(src, 4, 4)
    ys, xs = np.nonzero(d)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (6, 18, 6, 18) and len(ys) == 13 * 13
    d2 = LO.dilate_rect(src, 2, 4)       # o in [-4, 0]
    ys, xs = np.nonzero(d2)
    assert (ys.min(), ys.max(), xs.min(), xs.max()) == (10, 14, 10, 14)


# ------------------------------------------------------------------------------------- HIP path (simulator / GPU)
def _cloud(seed, n, W, H):
    """n LiDAR-like points: the synthetic room's surface samples, re-drawn with 1 cm range noise beyond its 26 000"""
    from oracle.refdrive.gen_golden_depth import synth_cloud
    base = synth_cloud(seed, min(n, 26000))
    if n <= len(base):
        return base
    rng = np.random.Generator(np.random.PCG64(seed + 1000))
    return base[rng.integers(0, len(base), n)] + rng.normal(0, 0.01, (n, 3))


def _pose(seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    q = np.array([1.0, 0, 0, 0]) + rng.normal(0, 0.05, 4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * y * y - 2 * z * z, 2 * x * y - 2 * z * w, 2 * x * z + 2 * y * w],
                  [2 * x * y + 2 * z * w, 1 - 2 * x * x - 2 * z * z, 2 * y * z - 2 * x * w],
                  [2 * x * z - 2 * y * w, 2 * y * z + 2 * x * w, 1 - 2 * x * x - 2 * y * y]])
    return R, rng.normal(0, 0.15, 3)


def check_golden(L):
    from holoagent_amd._lib import lidar_depth
    z, W, H = _gold()
    K = z["K"]
    for tag, scale in (("s1", 1), ("s2", 2)):
        uvz = np.vstack((z[tag + "_points_image"][0], z[tag + "_points_image"][1], z[tag + "_points_camera"][2])).T
        depth, stats, state, _ = lidar_depth([uvz], None, K, W, H, voxel_size=0, image_scale=scale, want_state=True, lib_=L)
        assert np.array_equal(depth[0], z[tag + "_depth"])
        assert np.array_equal(state == 1, z[tag + "_flags"]) and not (state == 2).any()
        assert stats[0].tolist() == [len(uvz), len(uvz), int((~z[tag + "_flags"]).sum()), int((z[tag + "_depth"] > 0).sum())]


def check_batch_against_oracle(L, W, H, n, frames, vs):
    """a batch of frames with their own clouds (one of them empty), projection + down-sampling included"""
    from holoagent_amd._lib import lidar_depth
    K = np.array([[0.73 * W, 0, W / 2 - 0.5], [0, 0.73 * W, H / 2 - 0.5], [0, 0, 1]])
    clouds, poses = [], []
    for f in range(frames):
        clouds.append(_cloud(20 + f, n, W, H) if f != 1 else np.zeros((0, 3)))
        R, t = _pose(40 + f)
        poses.append(np.concatenate([R, t[:, None]], axis=1))
    depth, stats, state, ms = lidar_depth(clouds, poses, K, W, H, voxel_size=vs, want_state=vs <= 0, lib_=L)
    for f in range(frames):
        want, flags, (pi, pc) = LO.lidar_depth_frame(clouds[f], poses[f][:, :3], poses[f][:, 3], K, W, H, voxel_size=vs)
        assert np.array_equal(depth[f], want), f
        assert stats[f, 1] == pi.shape[1] and stats[f, 2] == int((~flags).sum()) and stats[f, 3] == int((want > 0).sum())
        if vs <= 0 and len(clouds[f]):
            st = state[sum(len(c) for c in clouds[:f]):][:len(clouds[f])]
            assert int((st != 2).sum()) == pi.shape[1] and np.array_equal(st[st != 2] == 1, flags)
    assert (depth[1] == 0).all() and (depth[0] > 0).sum() > 1000
    return ms


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_simulator_equals_reference_run():
    from holoagent_amd._lib import HmsgLib
    check_golden(HmsgLib(PC.EMU_PATH))


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
@pytest.mark.parametrize("vs", [0.0, 0.02])
def test_simulator_batch_equals_oracle(vs):
    from holoagent_amd._lib import HmsgLib
    check_batch_against_oracle(HmsgLib(PC.EMU_PATH), 96, 72, 9000, 3, vs)


def test_bad_arguments_are_rejected():
    from holoagent_amd._lib import HmsgError, HmsgLib, lidar_depth
    if not os.path.exists(PC.EMU_PATH):
        pytest.skip("kernel simulator not built")
    L = HmsgLib(PC.EMU_PATH)
    with pytest.raises(HmsgError):       # per-point states and down-sampling do not go together
        lidar_depth([np.zeros((4, 3))], None, np.eye(3), 8, 8, voxel_size=0.02, want_state=True, lib_=L)
    with pytest.raises(HmsgError):
        lidar_depth([np.zeros((4, 3))], None, np.eye(3), 0, 8, voxel_size=0, lib_=L)
    depth, stats, _, _ = lidar_depth([], None, np.eye(3), 8, 8, voxel_size=0, lib_=L)
    assert depth.shape == (0, 8, 8)


@pytest.mark.gpu
def test_gpu_equals_reference_run():
    from holoagent_amd._lib import lib
    check_golden(lib())


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,n,frames,vs", [(96, 72, 9000, 3, 0.0), (320, 240, 120000, 4, 0.02), (640, 480, 150000, 2, 0.0)])
def test_gpu_batch_equals_oracle(W, H, n, frames, vs):
    from holoagent_amd._lib import lib
    check_batch_against_oracle(lib(), W, H, n, frames, vs)


# ------------------------------------------------------------------------- host mirror of the script (file formats)
def _write_pcd(path, pts, binary):
    with open(path, "wb") as f:
        hdr = ("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z intensity\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
               f"WIDTH {len(pts)}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS {len(pts)}\nDATA {'binary' if binary else 'ascii'}\n")
        f.write(hdr.encode())
        rec = np.concatenate([pts.astype(np.float32), np.ones((len(pts), 1), np.float32)], axis=1)
        if binary:
            f.write(rec.tobytes())
        else:
            for r in rec:
                f.write((" ".join(repr(float(v)) for v in r) + "\n").encode())


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_script_mirror_writes_the_oracles_images(tmp_path):
    from PIL import Image

    from holoagent_amd import lidar_depth as LD
    from holoagent_amd._lib import HmsgLib
    W, H = 96, 72
    kf_xyz = np.array([[0.0, 0, 0], [1.0, 0, 0.5], [30.0, 0, 0]])          # the third key frame is out of reach
    kf_dir = tmp_path / "PCD"
    kf_dir.mkdir()
    clouds = [_cloud(60 + k, 5000, W, H).astype(np.float32).astype(np.float64) for k in range(3)]
    with open(kf_dir / "scans_pos.txt", "w") as f:
        for k, c in enumerate(kf_xyz):
            f.write(f"{c[0]} {c[1]} {c[2]} 1 0 0 0\n")
            _write_pcd(kf_dir / f"{k}.pcd", clouds[k], binary=(k != 1))
    (tmp_path / "cameras.txt").write_text(f"# comment\n1 PINHOLE {W} {H} 70.0 70.0 47.5 35.5\n")
    quats = [(0.0, 0.0, 0.0, 1.0), (0.02, -0.05, 0.01, 0.998)]
    with open(tmp_path / "poses.txt", "w") as f:
        for i, (qx, qy, qz, qw) in enumerate(quats):
            f.write(f"{100.5 + i} {0.1 * i} 0.05 0.2 {qx} {qy} {qz} {qw}\n")
    rep = LD.generate_depth_images(str(tmp_path / "poses.txt"), str(tmp_path / "cameras.txt"), str(kf_dir),
                                   str(kf_dir / "scans_pos.txt"), str(tmp_path / "depth"), batch_frames=2,
                                   lib_=HmsgLib(PC.EMU_PATH))
    assert sorted(rep) == ["100.5000", "101.5000"]
    K, _, _ = LD.read_camera_intrinsics(str(tmp_path / "cameras.txt"))
    local_map = np.vstack([clouds[0], clouds[1]])                            # key frames within 4 m, in file order
    for ts, pose in LD.read_image_tum_trajectories(str(tmp_path / "poses.txt")).items():
        want, _, _ = LO.lidar_depth_frame(local_map, pose[:3, :3], pose[:3, 3], K, W, H, voxel_size=0.02)
        got = np.array(Image.open(tmp_path / "depth" / f"{ts}.png"))
        assert got.dtype == np.uint16 and np.array_equal(got, want) and (want > 0).sum() > 500
