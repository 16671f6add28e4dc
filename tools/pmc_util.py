#!/usr/bin/env python3
"""Aggregate tools/pmc_util.sh's counter passes into profiles/r01_pmc_util.json (per-kernel averages over launches)."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import BENCH_SOURCES, stamp  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r04"

NAMES = {"gemm_h256_kernel": "gemm_bf16_mfma", "gemm_w4_pair_kernel": "gemm_bf16_mfma_pair", "gemm_w4_kernel": "gemm_bf16_mfma", "attn_fwd_v3_kernel": "attn_fwd_mfma", "attn_fwd_w4_kernel": "attn_fwd_mfma", "attn_bwd_dkv_w4_kernel": "attn_bwd_dkv_mfma", "attn_bwd_dkv_v4_kernel": "attn_bwd_dkv_mfma",
         "attn_bwd_dq_v2_kernel": "attn_bwd_dq_mfma", "attn_bwd_dq_ds_kernel": "attn_bwd_dq_mfma"}
N_XCD, N_SIMD = 8, 1024

acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for d in ("pu1", "pu2"):
    for f in glob.glob(f"gpurun_out/{d}/*counter_collection.csv"):
        for row in csv.DictReader(open(f)):
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    acc[v][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/{d}/*kernel_trace.csv"):
        for row in csv.DictReader(open(f)):
            for k, v in NAMES.items():
                if k in row["Kernel_Name"]:
                    dur[v].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-9)

out = {"_note": "rocprofv3 --kernel-trace --pmc (two SQ passes) over `bench.py --steps 3 --warmup 1` on MI355X (tools/pmc_util.sh). SQ_* cycle "
                "counters are in quad-cycles summed over waves / SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs. mfma_util = "
                "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs) (the counter_defs.yaml MfmaUtil expression); wave-cycle "
                "shares = counter / SQ_WAVE_CYCLES; clock_ghz = GRBM_GUI_ACTIVE / 8 / kernel duration under the profiler (reads high on "
                "launches shorter than ~0.3 ms)."}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    t = sum(dur[k]) / len(dur[k])
    o = {"avg_duration_ms_profiled": t * 1e3, "launches_sampled": len(dur[k]) // 2, "counters": m}
    if "GRBM_GUI_ACTIVE" in m:
        cyc = m["GRBM_GUI_ACTIVE"] / N_XCD
        o["clock_ghz"] = cyc / t / 1e9
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            o["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD)
    if "SQ_WAVE_CYCLES" in m:
        for c, n in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt_or_barrier")):
            if c in m:
                o[n + "_share"] = m[c] / m["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_MFMA" in m and "SQ_INSTS_VALU" in m:
        o["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
    out[k] = o
out.update(stamp(BENCH_SOURCES))  # bench.py quotes this file only for the device sources it was measured on
json.dump(out, open(f"profiles/{ROUND}_pmc_util.json", "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters"} for k, v in out.items() if isinstance(v, dict)}, indent=1))
