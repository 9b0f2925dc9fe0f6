#!/bin/bash
# per-kernel statistics (rocprofv3 --kernel-trace --stats) of one 1000-frame step: the default line (batch fold) and the same with
# the incremental fold from the first frame (HMSG_FOLD_INCREMENTAL=1).   bash scripts/gpu_foldprof.sh <tag>
set -u
OUT=/root/repo/gpurun_out/${1:-foldprof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in batch incremental; do
  rm -rf /tmp/prof_$mode
  if [ $mode = incremental ]; then export HMSG_FOLD_INCREMENTAL=1; else unset HMSG_FOLD_INCREMENTAL; fi
  HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  cp /tmp/prof_$mode/*/*kernel_stats.csv $OUT/kernel_stats_$mode.csv
  grep "hmsg merge\]\|hmsg fold\] pairs" $OUT/bench_$mode.err | tail -n 2
  python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats_$mode.csv')))
tot = 0.0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if n.startswith(('k_db_', 'k_ov_', 'k_f_', 'k_ix_', 'k_publish', 'k_scan', 'k_concat', '__amd_rocclr_copy')):
        tot += float(r['TotalDurationNs'])
        print("%-26s calls %6s total %8.1f ms avg %8.1f us" % (n[:26], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print("sum of the fold's kernels and copies: %.1f ms" % (tot / 1e6))
PY
done
