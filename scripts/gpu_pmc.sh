#!/bin/bash
# the two PMC passes + the kernel statistics of the default line without its extra legs
set -u
OUT=/root/repo/gpurun_out/${1:-pmc}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python /root/repo/bench.py --steps 2 --warmup 0 --no-extras > $OUT/bench_under_rocprof.json 2>/dev/null
cp /tmp/prof_stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/prof_$c -- python /root/repo/bench.py --steps 1 --warmup 0 --no-extras > /dev/null 2>$OUT/pmc_$c.err
  python /root/repo/scripts/pmc_summary.py /tmp/prof_$c/*/*counter_collection.csv $OUT/pmc_$c.json
done
python /root/repo/scripts/pmc_combine.py $OUT/pmc_FETCH_SIZE.json $OUT/pmc_WRITE_SIZE.json $OUT/pmc_traffic.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over python bench.py --steps 1 --warmup 0 --no-extras; counters in KB, FETCH_SIZE doubled (gfx950)"
python -c "
import json; p=json.load(open('$OUT/pmc_traffic.json'))['kernels']
for k in ('k_ov_query','k_db_union','k_db_fill','k_db_cell','k_pool_gram','k_accum_ordered','k_slots','k_room_nn'): print(k, p.get(k))"
