"""Combine the two PMC passes (FETCH_SIZE, WRITE_SIZE; scripts/pmc_summary.py outputs) into profiles/rNN_pmc_traffic.json.
Counters are in KB; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-B requests tallied at 64 B)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from csrc_sha import csrc_sha16

fetch, write, out, note = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3], sys.argv[4]
res = {}
for k in sorted(set(fetch) | set(write)):
    f = fetch.get(k, {}).get("FETCH_SIZE")
    w = write.get(k, {}).get("WRITE_SIZE")
    if not f and not w:
        continue
    fb = f["mean"] * 1024 * 2 if f else 0.0
    wb = w["mean"] * 1024 if w else 0.0
    res[k] = dict(dispatches=(f or w)["dispatches"], fetch_bytes_per_launch_corrected=fb, write_bytes_per_launch=wb,
                  hbm_bytes_per_launch=fb + wb)
json.dump(dict(note=note, csrc_sha16=csrc_sha16(), kernels=res), open(out, "w"), indent=1)
print("kernels:", len(res))
