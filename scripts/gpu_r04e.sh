#!/bin/bash
# round 4: the new -m gpu tests (RCCL behind the C ABI with one rank, encoder hand-off, driver / rank_goal_views), then the
# per-kernel statistics of the default line (whole graph) under rocprofv3
set -u
OUT=/root/repo/gpurun_out/${1:-r04e}
mkdir -p $OUT
cd /root/repo
timeout 900 python -m pytest tests/test_comm_cabi.py tests/test_encoder_handoff.py tests/test_rooms_golden.py tests/test_query_driver_golden.py "tests/test_gpu_configs.py::test_bench_rccl_path_with_one_rank" tests/test_rooms_from_frames.py -m gpu -x -q > $OUT/gpu_pytest_new.log 2>&1
tail -n 6 $OUT/gpu_pytest_new.log
cd /tmp && export TMPDIR=/tmp
HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_def -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 > $OUT/bench_under_rocprof.json 2> $OUT/bench_under_rocprof.err
cp /tmp/prof_def/*/*kernel_stats.csv $OUT/kernel_stats.csv
grep "hmsg rooms\|hmsg room clouds\|hmsg merge\]" $OUT/bench_under_rocprof.err | tail -n 6
python - <<PY
import csv, json
rows = list(csv.DictReader(open('$OUT/kernel_stats.csv')))
for r in rows[:48]:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    print("%-26s calls %6s total %8.1f ms avg %9.1f us" % (n[:26], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
d = json.load(open('$OUT/bench_under_rocprof.json'))
print(d['value'], d['stage_ms_per_step'])
print(d['encoder_handoff'])
PY
