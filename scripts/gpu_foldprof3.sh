#!/bin/bash
# batch fold, per-kernel statistics of one 1000-frame step: flags / scan / scatter as three launches (HMSG_DB_COMPACT_SPLIT=1)
# against the one-launch compaction; then the fold-equality tests and a default line.   bash scripts/gpu_foldprof3.sh <tag>
set -u
OUT=/root/repo/gpurun_out/${1:-foldprof3}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for mode in ${MODES:-split fused}; do
  rm -rf /tmp/prof_$mode
  if [ $mode = split ]; then export HMSG_DB_COMPACT_SPLIT=1; else unset HMSG_DB_COMPACT_SPLIT; fi
  HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$mode -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 --no-extras > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  cp /tmp/prof_$mode/*/*kernel_stats.csv $OUT/kernel_stats_$mode.csv
  echo "== $mode"
  grep "hmsg merge\]" $OUT/bench_$mode.err | tail -n 1
  python - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/kernel_stats_$mode.csv')))
tot = 0.0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').split('(')[0]
    if n.startswith(('k_db_', 'k_ov_', 'k_publish', 'k_scan', 'k_concat', 'k_upload', '__amd_rocclr_copy')):
        tot += float(r['TotalDurationNs'])
        print("%-26s calls %6s total %8.1f ms avg %8.1f us" % (n[:26], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print("sum of the fold's kernels and copies: %.1f ms" % (tot / 1e6))
PY
done
unset HMSG_DB_COMPACT_SPLIT
cd /root/repo
timeout 900 python -m pytest tests/test_fold_pipeline.py tests/test_gpu_parity.py tests/test_fold_incremental.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -n 5 | tee $OUT/pytest.log
for i in 1 2; do python bench.py --no-extras --cpu-frames 0 --steps 3 2>/dev/null | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('stage_ms'))"; done | tee $OUT/bench_lines.txt
