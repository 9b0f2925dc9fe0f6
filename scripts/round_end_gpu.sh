#!/bin/bash
# Last GPU call of a round: what the driver will run on a fresh box -- the whole -m gpu suite, smoke(), the default bench line --
# on the final build, so that nothing meets the MI355X for the first time in the driver's hands.
#   gpurun --timeout 1700 -- 'bash scripts/round_end_gpu.sh r04z'
set -u
OUT=/root/repo/gpurun_out/${1:-end}
mkdir -p $OUT
cd /root/repo
timeout 1100 python -m pytest tests -m gpu -x -q > $OUT/gpu_pytest.log 2>&1
tail -n 4 $OUT/gpu_pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python -c "
import json
d = json.loads([l for l in open('$OUT/bench_default.json').read().splitlines() if l.startswith('{')][-1])
print(d['value'], d['ms_per_step'], d['stage_ms_per_step'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['seconds'], d['speedup_vs_cpu'], '| in flight', d['scenes_in_flight'], '| handed in', d['rooms_handed_in'])
print(d['roofline']); print(d['graph_counts'], d['queries_per_sec'], d['retrieval_room_stage_hit_rate'])"
