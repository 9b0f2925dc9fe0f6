#!/bin/bash
# Round profile set on the GPU box (run through gpurun from the repo root): the default bench line, the rocprofv3 kernel
# statistics of the same workload, the two PMC passes (FETCH_SIZE / WRITE_SIZE, each on its own with --kernel-trace
# only, as MI355X_MICROARCH.md prescribes), the D = 1024 line and the 1-rank RCCL path.  Everything lands in gpurun_out/$1/.
set -u
OUT=/root/repo/gpurun_out/${1:-prof}
mkdir -p $OUT
cd /root/repo
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err   # (the whole graph in the step; cpu_baseline = configs[0] in full: ~100 s)
python bench.py --feat-dim 1024 --cpu-frames 0 > $OUT/bench_d1024.json 2>/dev/null
HMSG_BENCH_FORCE_DIST=1 python bench.py --cpu-frames 0 > $OUT/bench_force_dist.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python /root/repo/bench.py --steps 2 --warmup 0 --cpu-frames 0 --inflight-steps 0 > $OUT/bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python /root/repo/bench.py --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 > /dev/null 2>$OUT/pmc_$c.err
  python /root/repo/scripts/pmc_summary.py /tmp/prof_$c/*/*counter_collection.csv $OUT/pmc_$c.json
done
python /root/repo/scripts/pmc_combine.py $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json $OUT/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0; counters in KB, FETCH_SIZE doubled (gfx950)"
ls -la $OUT
