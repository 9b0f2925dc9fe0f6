#!/bin/bash
# round 4, second GPU call: the room level from frames on the MI355X (new -m gpu tests), the default bench line (whole graph,
# room level beside the fusion) with the host-side assembly profile
set -u
OUT=/root/repo/gpurun_out/${1:-r04b}
mkdir -p $OUT
cd /root/repo
timeout 600 python -m pytest tests/test_rooms_from_frames.py tests/test_object_views.py tests/test_rooms_golden.py tests/test_gpu_parity.py -m gpu -x -q > $OUT/gpu_pytest_rooms.log 2>&1
tail -n 5 $OUT/gpu_pytest_rooms.log
HMSG_BENCH_PROFILE_ASSEMBLE=1 timeout 400 python bench.py --cpu-frames 0 --inflight-steps 0 > $OUT/bench_default.json 2> $OUT/bench_default.err
grep -v "^\[" $OUT/bench_default.err | head -n 60
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['graph_counts'], d['stage_ms_per_step'])"
