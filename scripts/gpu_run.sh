#!/bin/bash
# One parameterised GPU-box script (through gpurun from the repo root) instead of a new one-off script per experiment:
#   gpurun --timeout 900 -- 'bash scripts/gpu_run.sh <tag> <task> [<task> ...]'
# Everything lands in gpurun_out/<tag>/.  Tasks:
#   foldprof[:ENV=1,ENV2=1]  rocprofv3 --kernel-trace --stats of ONE 1000-frame step (rooms handed in, no extras) with the given
#                            environment switches; prints the fold's kernels (k_db_*, k_ov_*, scans, uploads, publishes) and their sum
#   foldtrace[:ENV=1,..]     rocprofv3 --kernel-trace of one scene -> scripts/fold_timeline.py: per-position kernel durations and launch gaps of a fold step
#   foldtime[:ENV=1,..]      HMSG_DEBUG_TIMING laps of the fold without the profiler (bench.py --rooms-handed-in --steps 2)
#   d1024                    the configs[2] line (D = 1024) with HMSG_DEBUG_TIMING laps of hmsg_query_hier
#   bench[:args]             python bench.py <args> -> bench_<n>.json (default args: --no-extras --cpu-frames 0 --steps 3)
#   tests[:pytest-args]      python -m pytest -m gpu -x -q <args> (default: tests)
#   smoke                    __graft_entry__.smoke()
#   stats                    rocprofv3 --kernel-trace --stats of bench.py --steps 2 --warmup 0 --no-extras -> kernel_stats.csv
#   pmc                      FETCH_SIZE / WRITE_SIZE passes (separate runs, --kernel-trace only) -> pmc_traffic.json
#   mfma                     SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES / SQ_INSTS_VALU_MFMA_MOPS_F32 / _F64 passes -> pmc_mfma.json
set -u
TAG=${1:-run}
shift
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
n_bench=0
fold_table() {   # $1 = kernel_stats csv
python - "$1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = 0.0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0].split('<')[0]
    if n.startswith(('k_db_', 'k_ov_', 'k_f_', 'k_ix_', 'k_publish', 'k_scan', 'k_concat', 'k_upload', 'k_uf_', '__amd_rocclr_copy')):
        tot += float(r['TotalDurationNs'])
        print("%-26s calls %6s total %8.1f ms avg %8.1f us" % (n[:26], r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3))
print("sum of the fold's kernels and copies: %.1f ms" % (tot / 1e6))
PY
}
for task in "$@"; do
  name=${task%%:*}
  arg=""
  [ "$task" != "$name" ] && arg=${task#*:}
  case $name in
    foldprof)
      label=$(echo "${arg:-default}" | tr -c 'A-Za-z0-9_\n' '_')
      ( cd /tmp
        for kv in $(echo "$arg" | tr ',' ' '); do export "$kv"; done
        rm -rf /tmp/prof_$label
        HMSG_DEBUG_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$label -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 --no-extras > $OUT/foldprof_$label.json 2> $OUT/foldprof_$label.err
        cp /tmp/prof_$label/*/*kernel_stats.csv $OUT/kernel_stats_fold_$label.csv )
      echo "== foldprof $arg"
      grep "hmsg merge\]" $OUT/foldprof_$label.err | tail -n 4
      fold_table $OUT/kernel_stats_fold_$label.csv
      ;;
    foldtrace)
      # per-dispatch kernel trace of one scene's fold -> scripts/fold_timeline.py (durations and gaps by position in the step)
      label=$(echo "${arg:-default}" | tr -c 'A-Za-z0-9_\n' '_')
      ( cd /tmp
        for kv in $(echo "$arg" | tr ',' ' '); do export "$kv"; done
        rm -rf /tmp/trace_$label
        rocprofv3 --kernel-trace --output-format csv -d /tmp/trace_$label -- python /root/repo/bench.py --rooms-handed-in --steps 1 --warmup 0 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 --no-extras > /dev/null 2> $OUT/foldtrace_$label.err
        python /root/repo/scripts/fold_timeline.py /tmp/trace_$label/*/*kernel_trace.csv > $OUT/fold_timeline_$label.txt )
      echo "== foldtrace $arg"
      cat $OUT/fold_timeline_$label.txt
      ;;
    foldtime)
      # host-side phase times of the fold WITHOUT the profiler (HMSG_DEBUG_TIMING laps), one scene, rooms handed in
      label=$(echo "${arg:-default}" | tr -c 'A-Za-z0-9_\n' '_')
      ( cd /root/repo
        for kv in $(echo "$arg" | tr ',' ' '); do export "$kv"; done
        HMSG_DEBUG_TIMING=1 python bench.py --rooms-handed-in --steps 2 --warmup 1 --cpu-frames 0 --inflight-steps 0 --encoder-frames 0 --no-extras > $OUT/foldtime_$label.json 2> $OUT/foldtime_$label.err )
      echo "== foldtime $arg"
      grep "hmsg merge\]\|hmsg fold\]" $OUT/foldtime_$label.err | tail -n 6
      python -c "
import json; d = json.loads([l for l in open('$OUT/foldtime_$label.json').read().splitlines() if l.startswith('{')][-1]); print('foldtime:', d['value'], d['stage_ms_per_step'])"
      ;;
    d1024)
      cd /root/repo
      HMSG_DEBUG_TIMING=1 python bench.py --feat-dim 1024 --cpu-frames 0 --no-extras --steps 2 --warmup 1 > $OUT/bench_d1024.json 2> $OUT/bench_d1024.err
      grep "query_hier" $OUT/bench_d1024.err | tail -n 8
      python -c "
import json; d = json.loads([l for l in open('$OUT/bench_d1024.json').read().splitlines() if l.startswith('{')][-1]); print('d1024:', d['value'], d['stage_ms_per_step'], d['queries_per_sec'])"
      ;;
    bench)
      cd /root/repo
      n_bench=$((n_bench + 1))
      python bench.py ${arg:---no-extras --cpu-frames 0 --steps 3} > $OUT/bench_$n_bench.json 2> $OUT/bench_$n_bench.err
      python -c "
import json; d = json.loads([l for l in open('$OUT/bench_$n_bench.json').read().splitlines() if l.startswith('{')][-1]); print('bench $n_bench:', d['value'], d['ms_per_step'], d['stage_ms_per_step'], d.get('roofline'))"
      ;;
    tests)
      cd /root/repo
      timeout 1200 python -m pytest -m gpu -x -q -p no:cacheprovider ${arg:-tests} > $OUT/gpu_pytest.log 2>&1
      tail -n 5 $OUT/gpu_pytest.log
      ;;
    smoke)
      cd /root/repo
      timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 | tee $OUT/smoke.log
      ;;
    stats)
      ( cd /tmp; rm -rf /tmp/prof_stats
        rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python /root/repo/bench.py --steps 2 --warmup 0 --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
        cp /tmp/prof_stats/*/*kernel_stats.csv $OUT/kernel_stats.csv )
      head -n 12 $OUT/kernel_stats.csv | cut -c1-150
      ;;
    pmc)
      ( cd /tmp
        for c in FETCH_SIZE WRITE_SIZE; do
          rm -rf /tmp/prof_$c
          rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python /root/repo/bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>$OUT/pmc_$c.err
          python /root/repo/scripts/pmc_summary.py /tmp/prof_$c/*/*counter_collection.csv $OUT/pmc_$c.json
        done )
      python /root/repo/scripts/pmc_combine.py $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json $OUT/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 1 --warmup 0 --no-extras; counters in KB, FETCH_SIZE doubled (gfx950)"
      ;;
    mfma)
      ( cd /tmp
        for c in SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE; do
          rm -rf /tmp/prof_$c
          rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python /root/repo/bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>$OUT/pmc_$c.err
          python /root/repo/scripts/pmc_summary.py /tmp/prof_$c/*/*counter_collection.csv $OUT/pmc_$c.json || true
        done )
      python /root/repo/scripts/pmc_mfma.py $OUT $OUT/pmc_mfma.json || true
      ;;
    *)
      echo "unknown task $task"
      ;;
  esac
done
ls $OUT | head -n 40
