"""MFMA utilisation of the two matrix-core kernels from rocprofv3 PMC passes (scripts/gpu_run.sh mfma: one counter per pass,
--kernel-trace only).  usage: pmc_mfma.py <dir with pmc_<COUNTER>.json> <out.json>

MfmaUtil is the gfx94x derived-counter formula (ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
    100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4 SIMDs)
and, beside it, the share of the shader-busy time SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CYCLES * CUs / SEs...) is NOT derived here:
SQ_BUSY_CYCLES is reported raw (its aggregation over shader engines differs between ROCm releases).  MOPS counters are in units of
512 FLOP-equivalents ("MOPS" = 512 ops) per the gfx9 counter descriptions; raw values are kept so the reader can re-derive."""
import json
import os
import sys

d, out = sys.argv[1], sys.argv[2]
CUS, SIMDS = 256, 4
counters = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_F32", "SQ_INSTS_VALU_MFMA_MOPS_F64", "GRBM_GUI_ACTIVE"]
data = {}
for c in counters:
    p = os.path.join(d, "pmc_%s.json" % c)
    if os.path.exists(p):
        data[c] = json.load(open(p))
res = {}
for kern in sorted({k for v in data.values() for k in v}):
    if not any(s in kern for s in ("k_pool_gram", "k_gemm_f64")):
        continue
    row = {}
    for c in counters:
        e = data.get(c, {}).get(kern, {}).get(c)
        if e:
            row[c] = dict(dispatches=e["dispatches"], mean_per_launch=e["mean"])
    busy = row.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean_per_launch")
    gui = row.get("GRBM_GUI_ACTIVE", {}).get("mean_per_launch")
    if busy and gui:
        row["MfmaUtil_percent_gfx94x_formula"] = 100.0 * busy / (gui * CUS * SIMDS)
    res[kern] = row
json.dump(dict(note="rocprofv3 --kernel-trace --pmc <one counter per pass> over python bench.py --steps 1 --warmup 0 --no-extras; "
                    "MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * 256 CUs * 4 SIMDs) (gfx94x derived formula)",
               kernels=res), open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:2000])
