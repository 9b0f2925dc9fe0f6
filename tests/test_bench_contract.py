"""bench.py end to end against the kernel simulator (HMSG_BENCH_EMU, tiny sizes): a benchmark line that crashes on the GPU box
is a round without a measurement, so every code path of a step -- including `--full-graph` (rooms from the device watershed,
room clouds, room embeddings, View nodes, the view <-> object test on the device, hierarchical retrieval over the segmented
rooms) -- and the cpu_baseline leg run here first.  The numbers mean nothing (the line says "emulated": true); the contract
of the line is what is checked."""
import json
import os
import subprocess
import sys

import pytest

from tests import parity_common as PC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.environ.get("HMSG_EMU_SLOW"), reason="minutes on the kernel simulator (HMSG_EMU_SLOW=1); its twin runs on the MI355X")
@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_bench_line_full_graph_on_the_simulator():
    env = dict(os.environ, HMSG_BENCH_EMU=PC.EMU_PATH)
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", "12", "--queries", "8", "--feat-dim", "16", "--width", "96",
           "--height", "72", "--scene-shape", "2,1,3.2,2.6,3.0,36,3", "--steps", "1", "--warmup", "0", "--cpu-frames", "2",
           "--inflight-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    d = json.loads(lines[-1])                                   # the JSON line is the last thing printed
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["emulated"] is True and d["full_graph"] is True
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["value"] - 12 / (d["ms_per_step"] / 1e3)) < 1e-2 * d["value"] + 1e-3
    assert "workload" in d["config"] and "model" not in d["config"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] > 0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-5
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and "sample" in cb
    gc = d["graph_counts"]
    assert gc["floors"] >= 1 and gc["rooms"] >= 1 and gc["views"] == 12 and gc["objects"] >= 3 and gc["view_object_edges"] >= 3
    assert "device watershed" in d["metric"] and "without views" not in d["metric"]
    st = d["stage_ms_per_step"]
    for k in ("add_frames", "finalize_map", "fuse_frames", "merge_instances", "pool_instances", "assemble_graph", "retrieval"):
        assert st[k] >= 0


@pytest.mark.skipif(not os.path.exists(PC.EMU_PATH), reason="kernel simulator not built")
def test_bench_two_ranks_time_the_same_step_on_the_simulator():
    """`bench.py --gpus 2` (scene per GPU) with two simulator ranks: gloo for the process group, the test-double librccl
    (tests/rccl_double) under the C-ABI collectives.  Every rank builds its WHOLE graph (the step one GPU times), the node tables are
    all-gathered together with the levels above them by hmsg_graph_allgather_index, and the ranks answer their share
    of the queries coarse to fine -- the line carries the same graph_counts as the one-GPU line, per rank."""
    from tests.test_comm_world import _double
    env = dict(os.environ, HMSG_BENCH_EMU=PC.EMU_PATH, HMSG_RCCL_LIB=_double(), MASTER_ADDR="127.0.0.1", MASTER_PORT="29631")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--frames", "12", "--queries", "8", "--feat-dim", "16", "--width", "96", "--height", "72",
           "--scene-shape", "2,1,3.2,2.6,3.0,36,3", "--steps", "1", "--warmup", "0", "--cpu-frames", "0", "--inflight-steps", "0"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["emulated"] is True and d["full_graph"] is True and d["scaling"] == "weak"
    assert d["graph_level"].startswith("C ABI graph object")
    assert "configs[3]" in d["config"]["workload"] and d["config"]["parallelism"] == "scene-per-gpu x2"
    per = d["graph_counts_all_ranks"]
    # (per rank: what hmsg_graph_allgather_index's offsets say -- floors, rooms, object nodes of every rank's graph)
    assert len(per) == 2 and all(c["floors"] >= 1 and c["rooms"] >= 1 and c["objects"] >= 3 for c in per)
    assert d["graph_counts"]["views"] == 12 and d["value"] > 0 and len(d["per_rank_frames_per_s"]) == 2
    st = d["stage_ms_per_step"]
    assert "room_level/device" in st and "assemble/graph_finish" in st and st["retrieval"] > 0
