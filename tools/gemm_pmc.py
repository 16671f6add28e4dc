#!/usr/bin/env python3
"""Aggregate tools/gemm_pmc.sh's per-case counter passes into profiles/rNN_gemm_pmc.json: per case the kernels that ran, their average
duration under the profiler, matrix-pipe utilisation, VALU per MFMA, wave-cycle shares, clock, and bytes beyond L2 per launch against
the algorithmic bytes."""
import collections
import csv
import glob
import json
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import stamp  # noqa: E402

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r06"
N_XCD, N_SIMD = 8, 1024
PEAK = {"bf16": 2500.0, "f32": 157.3}
FLOP_PER_MFMA = {"bf16": 32 * 32 * 16 * 2 * 1.0, "f32": 32 * 32 * 2 * 2 * 1.0}
SKIP = 4   # the first launches of a process run on a cold clock


def rows(pattern):
    for f in glob.glob(pattern):
        yield from csv.DictReader(open(f))


out = {"_note": "rocprofv3 --kernel-trace --pmc, one process per case and pass (tools/gemm_pmc.sh -> tools/gemm_pmc_case.py: 4 + 6 launches, the "
                "first 4 dropped), MI355X. mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); clock_ghz = GRBM_GUI_ACTIVE / 8 / "
                "duration; bytes beyond L2 = (2 FETCH_SIZE + WRITE_SIZE) KB (the guide's gfx950 correction); tflops = 2 n^3 / the profiled duration "
                "(the profiler's serialisation costs a few per cent against tools/gemm_sweep.py's HIP-event figures)."}
for line in open("gpurun_out/gp/index.txt"):
    i, case = line.strip().split("|")
    dt, n, lay = case.split()
    n = int(n)
    per = collections.defaultdict(lambda: collections.defaultdict(list))   # kernel -> counter -> values
    dur = collections.defaultdict(list)
    for p in (1, 2, 3, 4):
        seen = collections.Counter()
        for r in sorted(rows(f"gpurun_out/gp/c{i}p{p}/*counter_collection.csv"), key=lambda r: int(r["Dispatch_Id"])):
            if "gemm" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kf::", "")
            seen[(k, r["Counter_Name"])] += 1
            if seen[(k, r["Counter_Name"])] > SKIP:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        seen = collections.Counter()
        for r in sorted(rows(f"gpurun_out/gp/c{i}p{p}/*kernel_trace.csv"), key=lambda r: int(r["Start_Timestamp"])):
            if "gemm" not in r["Kernel_Name"]:
                continue
            k = r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", "").replace("kf::", "")
            seen[k] += 1
            if seen[k] > SKIP:
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    o = {"kernels": {}}
    total_t = 0.0
    for k, d in per.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        t = sum(dur[k]) / max(len(dur[k]), 1)
        total_t += t
        ko = {"avg_duration_ms_profiled": t * 1e3, "counters": m}
        if "GRBM_GUI_ACTIVE" in m and t > 0:
            cyc = m["GRBM_GUI_ACTIVE"] / N_XCD
            ko["clock_ghz"] = cyc / t / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
                ko["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * N_SIMD)
        if "SQ_WAVE_CYCLES" in m:
            for c, nm in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt_or_barrier")):
                if c in m:
                    ko[nm + "_share"] = m[c] / m["SQ_WAVE_CYCLES"]
        if m.get("SQ_INSTS_MFMA"):
            ko["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
            ko["mfma_flops_executed"] = m["SQ_INSTS_MFMA"] * FLOP_PER_MFMA[dt]
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            ko["bytes_beyond_l2"] = (2 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024
        o["kernels"][k] = ko
    es = 2 if dt == "bf16" else 4
    alg = 3 * n * n * es + (n * n * es + n * es if lay.endswith("+epi") else 0)
    o["algorithmic_bytes"] = alg
    o["bytes_beyond_l2"] = sum(k.get("bytes_beyond_l2", 0) for k in o["kernels"].values())
    o["traffic_over_algorithmic"] = o["bytes_beyond_l2"] / alg if alg else None
    if total_t > 0:
        o["ms_profiled"] = total_t * 1e3
        o["tflops_profiled"] = 2.0 * n ** 3 / total_t / 1e12
        o["frac_of_mfma_peak"] = o["tflops_profiled"] / PEAK[dt]
    out[case] = o
out.update(stamp(("gemm.hip", "common.h", "runtime.hip")))
json.dump(out, open(f"profiles/{ROUND}_gemm_pmc.json", "w"), indent=1)
for case, o in out.items():
    if isinstance(o, dict) and "kernels" in o:
        main = max(o["kernels"].items(), key=lambda kv: kv[1]["avg_duration_ms_profiled"], default=(None, {}))
        print(f"{case:20s} {o.get('ms_profiled', 0):8.4f} ms {o.get('tflops_profiled', 0):7.1f} TF/s  mfma_util {main[1].get('mfma_util', 0):.3f}  valu/mfma {main[1].get('valu_per_mfma', 0):.2f}  "
              f"clock {main[1].get('clock_ghz', 0):.2f} GHz  traffic x{o.get('traffic_over_algorithmic') or 0:.2f}  {main[0]}")
