# Matrix-pipe utilisation, VALU per MFMA, wave-cycle shares and clock of the head-size-64 attention kernels (the generated streams): two
# rocprofv3 --pmc passes over `tools/attn_bench.py --B 8 --H 64 --D 64`, aggregated like tools/pmc_util.sh -> gpurun_out/$R_attn_d64_pmc.json
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r06}"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pd$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pd$i -o r -- python3 tools/attn_bench.py --B 8 --H 64 --D 64 --rounds 3 > gpurun_out/pd$i.log 2>&1
done
python3 - "$R" <<'PY'
import collections, csv, glob, json, sys
sys.path.insert(0, ".")
from bench import BENCH_SOURCES, stamp
R = sys.argv[1]
NAMES = {"attn_fwd_w4_kernel": "attn_fwd_mfma_d64", "attn_bwd_dkv_w4_kernel": "attn_bwd_dkv_mfma_d64", "attn_bwd_dq_ds_kernel": "attn_bwd_dq_mfma_d64"}
acc, dur = collections.defaultdict(lambda: collections.defaultdict(list)), collections.defaultdict(list)
for d in ("pd1", "pd2"):
    for f in glob.glob(f"gpurun_out/{d}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            for k, v in NAMES.items():
                if k in r["Kernel_Name"]:
                    acc[v][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(f"gpurun_out/{d}/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            for k, v in NAMES.items():
                if k in r["Kernel_Name"]:
                    dur[v].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
out = {"_note": "rocprofv3 --kernel-trace --pmc over tools/attn_bench.py --B 8 --H 64 --D 64 (tools/pmc_d64.sh); definitions as in r05_pmc_util.json"}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    t = sum(dur[k]) / len(dur[k])
    cyc = m["GRBM_GUI_ACTIVE"] / 8
    o = {"avg_duration_ms_profiled": t * 1e3, "clock_ghz": cyc / t / 1e9}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m: o["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024)
    if "SQ_WAVE_CYCLES" in m:
        for c, n in (("SQ_ACTIVE_INST_ANY", "issuing"), ("SQ_WAIT_INST_ANY", "issue_stall"), ("SQ_WAIT_ANY", "waitcnt_or_barrier")):
            o[n + "_share"] = m[c] / m["SQ_WAVE_CYCLES"]
    if m.get("SQ_INSTS_MFMA"): o["valu_per_mfma"] = (m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"]
    out[k] = o
out.update(stamp(BENCH_SOURCES))
json.dump(out, open(f"gpurun_out/{R}_attn_d64_pmc.json", "w"), indent=1)
print(json.dumps({k: {a: round(b, 3) for a, b in v.items()} for k, v in out.items() if isinstance(v, dict) and "clock_ghz" in v}, indent=1))
PY
