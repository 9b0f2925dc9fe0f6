#!/bin/bash
# round 4, third GPU call: where the room level's device stage spends its 156 ms, hot-word microbenchmark, and the per-kernel
# statistics of the INCREMENTAL fold at configs[1] (HMSG_FOLD_INCREMENTAL=1) next to the batch fold's
set -u
OUT=/root/repo/gpurun_out/${1:-r04c}
mkdir -p $OUT
cd /root/repo
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/atom_bench scripts/microbench/atom_bench.hip && timeout 120 /tmp/atom_bench > $OUT/atom_bench.txt 2>&1
cat $OUT/atom_bench.txt
HMSG_DEBUG_TIMING=1 timeout 300 python bench.py --cpu-frames 0 --inflight-steps 0 --steps 1 --warmup 1 > $OUT/bench_timing.json 2> $OUT/bench_timing.err
grep "hmsg rooms\|hmsg merge\|hmsg fold" $OUT/bench_timing.err | tail -n 12
cd /tmp && export TMPDIR=/tmp
HMSG_FOLD_INCREMENTAL=1 HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_inc -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 > $OUT/bench_incremental_under_rocprof.json 2> $OUT/bench_incremental.err
cp /tmp/prof_inc/*/*kernel_stats.csv $OUT/kernel_stats_incremental.csv
grep "hmsg merge\|hmsg fold" $OUT/bench_incremental.err | tail -n 12
head -n 45 $OUT/kernel_stats_incremental.csv | cut -c1-150
