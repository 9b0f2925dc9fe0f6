"""MFMA utilisation of the two matrix-core kernels from rocprofv3 PMC passes (scripts/gpu_run.sh mfma: one counter per pass,
--kernel-trace only).  usage: pmc_mfma.py <dir with pmc_<COUNTER>.json> <out.json>

MfmaUtil is the gfx94x derived-counter formula (ROCm 7.2 ships no gfx950 section, MI355X_MICROARCH.md "rocprofv3 PMC slots"):
    100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4 SIMDs)
with one correction for this chip: rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs (for k_pool_gram it is 8.6 x the kernel's
duration x 2.1 GHz, and SQ_INSTS_VALU_MFMA_MOPS_F32 x 512 FLOP reproduces the kernel's FLOP count to 2 %), so the elapsed cycles
of ONE clock domain are GRBM_GUI_ACTIVE / 8:
    MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
The uncorrected gfx94x figure is kept beside it.  SQ_BUSY_CYCLES and the MOPS counters are reported raw."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import os
import sys

d, out = sys.argv[1], sys.argv[2]
CUS, SIMDS, XCDS = 256, 4, 8
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
        row["MfmaUtil_percent"] = 100.0 * busy / (gui / XCDS * CUS * SIMDS)
    res[kern] = row
json.dump(dict(csrc_sha16=__import__('csrc_sha').csrc_sha16(), note="rocprofv3 --kernel-trace --pmc <one counter per pass> over python bench.py --steps 1 --warmup 0 --no-extras; "
                    "MfmaUtil = 100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 256 CUs * 4 SIMDs); the uncorrected gfx94x formula beside it",
               kernels=res), open(out, "w"), indent=1)
print(json.dumps(res, indent=1)[:2000])
