#!/bin/bash
set -u
OUT=/root/repo/gpurun_out/${1:-r04i}
mkdir -p $OUT
cd /root/repo
timeout 300 python -m pytest tests/test_rooms_from_frames.py tests/test_rooms_golden.py -m gpu -x -q 2>&1 | tail -n 2
for mode in before beside; do
  if [ $mode = beside ]; then export HMSG_BENCH_ROOMS_BESIDE_FOLD=1; else unset HMSG_BENCH_ROOMS_BESIDE_FOLD; fi
  HMSG_DEBUG_TIMING=1 timeout 400 python bench.py --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  grep "hmsg rooms\|hmsg room clouds" $OUT/bench_$mode.err | tail -n 2
  python -c "
import json; d = json.load(open('$OUT/bench_$mode.json')); print('$mode', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
