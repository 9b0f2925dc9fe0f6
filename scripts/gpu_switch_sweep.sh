#!/bin/bash
# the 10 000-frame 1280x720 streamed episode under different switch points of the merge fold (batch -> incremental:
# HMSG_FOLD_SWITCH = running mean of the batch size in points).   bash scripts/gpu_switch_sweep.sh <tag>
set -u
OUT=/root/repo/gpurun_out/${1:-switch}
mkdir -p $OUT
cd /root/repo
for sw in ${SWEEP:-600000 1000000 1500000}; do
  HMSG_FOLD_SWITCH=$sw timeout 300 python scripts/bench_stream_episode.py --frames 10000 --chunk 100 --no-feats-check > $OUT/episode_$sw.json 2> $OUT/episode_$sw.err
  python - <<PY
import json
d = json.loads([l for l in open('$OUT/episode_$sw.json').read().splitlines() if l.startswith('{')][-1])
print('$sw', d.get('value'), d.get('seconds'), {k: v for k, v in d.get('stage_s', d.get('stages', {})).items()} if isinstance(d.get('stage_s', d.get('stages')), dict) else list(d.keys())[:12])
PY
done
