#!/bin/bash
# round 4: whole -m gpu suite + the default line (timing lines of the room level and the merge on stderr)
set -u
OUT=/root/repo/gpurun_out/${1:-r04d}
mkdir -p $OUT
cd /root/repo
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/gpu_pytest.log 2>&1
tail -n 4 $OUT/gpu_pytest.log
HMSG_DEBUG_TIMING=1 timeout 400 python bench.py --cpu-frames 0 --inflight-steps 0 > $OUT/bench_default.json 2> $OUT/bench_default.err
grep "hmsg rooms\|hmsg merge\]" $OUT/bench_default.err | tail -n 6
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['graph_counts'], d['stage_ms_per_step']); print(d['kernels_ms_last_step'])"
