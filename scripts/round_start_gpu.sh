#!/bin/bash
# First GPU call of a round (through gpurun from the repo root): everything that was written or changed after the previous
# round's GPU minutes ran out gets its first run on the MI355X here, then the numbers the next design decision hangs on.
#   gpurun --timeout 1500 -- 'bash scripts/round_start_gpu.sh r04a'
# 1. the whole -m gpu suite with HMSG_TEST_UNVALIDATED=1 (tests of code that has only run on the kernel simulator so far are
#    skipped without it, so that the driver's round-end run never meets code for the first time);
# 2. scripts/microbench/step_bench.hip (grid barriers, phase floor, cross-workgroup hand-over: DESIGN.md section 7);
# 3. smoke(), the default bench line, and `bench.py --full-graph` (A9 + the A10 view test inside the timed step: only run on
#    the simulator so far, tests/test_bench_contract.py).
set -u
OUT=/root/repo/gpurun_out/${1:-start}
mkdir -p $OUT
cd /root/repo
HMSG_TEST_UNVALIDATED=1 timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/gpu_pytest.log 2>&1
tail -n 5 $OUT/gpu_pytest.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/step_bench scripts/microbench/step_bench.hip && timeout 120 /tmp/step_bench > $OUT/step_bench.txt 2>&1
cat $OUT/step_bench.txt
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 400 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 python bench.py --full-graph --cpu-frames 0 --inflight-steps 0 > $OUT/bench_full_graph.json 2> $OUT/bench_full_graph.err
python -c "
import json; d = json.load(open('$OUT/bench_full_graph.json')); print('full graph:', d['value'], d['graph_counts'], d['stage_ms_per_step'])"
python -c "
import json; d = json.load(open('$OUT/bench_default.json')); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
