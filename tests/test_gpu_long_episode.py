"""BASELINE.json configs[4] at its single-GPU size inside the -m gpu suite: a 10 000-frame 1280 x 720 episode streamed through
the C ABI in chunks (scripts/bench_stream_episode.py: 157 GB of resident frame store, the incremental fold taking over from
the batch fold, the fold's arenas carved out of the handed-back frame store), checked through properties that do not
depend on the size: every instance keeps at least min_instance_points points and lies inside the map's box (a 3-D mask
point is the mean of map points of one voxel, an instance is a subset of mask points), a voxel is counted at most once
per frame, (nearly) every voxel was seen, and object queries land in the queried object's room."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.gpu
def test_streamed_10k_frame_720p_episode_properties():
    import torch
    free, total = torch.cuda.mem_get_info(0)
    if total < 250e9:
        pytest.skip("needs the MI355X's 288 GB (157 GB frame store + the fold's arenas)")
    F = int(os.environ.get("HMSG_TEST_LONG_FRAMES", "10000"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_stream_episode.py"), "--frames", str(F), "--chunk", "100",
                        "--queries", "200"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    p = d["properties"]
    print(json.dumps(d))
    assert d["frames"] == F and d["image"] == [1280, 720]
    assert d["instances"] >= 100 and d["objects"] >= 100 and d["map_voxels"] > 50000
    assert p["min_instance_points"] >= 10
    assert p["boxes_inside_map"] is True
    assert 1.0 <= p["counter_max"] <= F and p["voxels_seen"] >= 0.9
    assert p["top1_in_the_queried_objects_room"] >= 0.7
