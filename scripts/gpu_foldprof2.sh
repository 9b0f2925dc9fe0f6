#!/bin/bash
# incremental fold with every component without a dominant anchor sent through the batch kernels (HMSG_FOLD_BIG_ACTIVE=<n>)
set -u
OUT=/root/repo/gpurun_out/${1:-foldprof2}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ba in 0 3000; do
  rm -rf /tmp/prof_$ba
  HMSG_FOLD_INCREMENTAL=1 HMSG_FOLD_BIG_ACTIVE=$ba HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$ba -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 > $OUT/bench_$ba.json 2> $OUT/bench_$ba.err
  cp /tmp/prof_$ba/*/*kernel_stats.csv $OUT/kernel_stats_$ba.csv
  echo "== big_active $ba"
  grep "hmsg fold\] pairs\|hmsg fold\] steps\|dbscan batches" $OUT/bench_$ba.err | tail -n 3
  python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats_$ba.csv')))
tot = 0.0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if n.startswith(('k_db_', 'k_ov_', 'k_f_', 'k_ix_', 'k_publish', 'k_scan', 'k_concat', '__amd_rocclr_copy')):
        tot += float(r['TotalDurationNs'])
        if float(r['TotalDurationNs']) > 3e6:
            print("%-26s calls %6s total %8.1f ms avg %8.1f us" % (n[:26], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print("sum of the fold's kernels and copies: %.1f ms" % (tot / 1e6))
PY
done
