#!/usr/bin/env python3
"""Aggregate the three rocprofv3 passes of tools/membound_prof.sh into gpurun_out/rNN_membound_rocprof.json (copy it to profiles/):
for every case of tools/membound_bench.py the rocprofv3-reported kernel time per call, the bytes beyond L2 per call
((2 x FETCH_SIZE + WRITE_SIZE) KiB: FETCH_SIZE reports half the bytes of wide coalesced reads on gfx950, MI355X_MICROARCH.md) and the two
bandwidth figures they give: ALGORITHMIC bytes / rocprof time (the roofline figure) and COUNTER bytes / rocprof time (what the memory
system actually moved). Cases are cut out of the dispatch list at the one-workgroup f64 fills membound_bench.py --markers launches."""
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from bench import MEMBOUND_SOURCES, stamp  # noqa: E402

R = sys.argv[1] if len(sys.argv) > 1 else "r04"
OUT = ROOT / "gpurun_out"


def is_marker(name, grid):  # the 64-element f64 fill: the 8-byte-unit instance of the same-dtype kernel (no case of the bench uses it), one workgroup
    return "ew_same_kernel<unsigned long" in name and grid <= 1024


def segments(rows, name_key, grid_of):
    """rows in dispatch order -> list of lists (one per case), split at the markers."""
    segs, cur = [], None
    for r in rows:
        if is_marker(r[name_key], grid_of(r)):
            cur = []
            segs.append(cur)
        elif cur is not None:
            cur.append(r)
    return segs


def main():
    bench = json.loads((OUT / f"{R}_membound.json").read_text())
    cases = sorted(bench, key=lambda k: bench[k]["case_index"])
    trace = sorted(csv.DictReader(open(glob.glob(str(OUT / "mbT" / "**" / "*kernel_trace.csv"), recursive=True)[0])), key=lambda r: int(r["Dispatch_Id"]))
    tsegs = segments(trace, "Kernel_Name", lambda r: int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))

    def counter(d, name):
        rows = [r for f in glob.glob(str(OUT / d / "**" / "*counter_collection.csv"), recursive=True) for r in csv.DictReader(open(f)) if r["Counter_Name"] == name]
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        return segments(rows, "Kernel_Name", lambda r: int(r["Grid_Size"]))
    fsegs, wsegs = counter("mbF", "FETCH_SIZE"), counter("mbW", "WRITE_SIZE")
    assert len(tsegs) == len(fsegs) == len(wsegs) == len(cases), (len(tsegs), len(fsegs), len(wsegs), len(cases))
    out = {"_note": "rocprofv3 --kernel-trace --stats (durations) and separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes over "
                    "`python3 tools/membound_bench.py --markers --rounds 5` on MI355X (tools/membound_prof.sh). bytes_beyond_l2 = "
                    "(2 FETCH_SIZE + WRITE_SIZE) KiB per call (gfx950 FETCH correction, MI355X_MICROARCH.md; Infinity-Cache hits are counted: an "
                    "upper bound on HBM traffic). GBps_algorithmic = algorithmic bytes / rocprof kernel time: the roofline figure, peak 8000 GB/s "
                    "(6300 measured copy ceiling)."}
    for name, ts, fs, ws in zip(cases, tsegs, fsegs, wsegs):
        rounds = bench[name]["rounds"]
        per = len(ts) // (rounds + 1)          # kernel launches per call (the first call of a case is its warm-up)
        use = ts[per:]
        ms = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in use) / 1e6 / rounds
        fk = sum(float(r["Counter_Value"]) for r in fs[per:]) / rounds
        wk = sum(float(r["Counter_Value"]) for r in ws[per:]) / rounds
        traffic = (2 * fk + wk) * 1024
        kern = sorted({r["Kernel_Name"].split("(")[0][:90] for r in use})
        out[name] = {"rocprof_ms_per_call": ms, "hip_event_ms_per_call": bench[name]["ms"], "kernel_launches_per_call": per,
                     "algorithmic_bytes": bench[name]["algorithmic_bytes"], "bytes_beyond_l2": traffic, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                     "GBps_algorithmic": bench[name]["algorithmic_bytes"] / (ms * 1e-3) / 1e9, "GBps_counters": traffic / (ms * 1e-3) / 1e9,
                     "frac_of_8TBps": bench[name]["algorithmic_bytes"] / (ms * 1e-3) / 1e9 / 8000.0,
                     "traffic_over_algorithmic": traffic / bench[name]["algorithmic_bytes"], "kernels": kern}
        print(f"{name:60s} {ms:8.4f} ms  alg {out[name]['GBps_algorithmic']:8.1f} GB/s  counters {out[name]['GBps_counters']:8.1f} GB/s  x{out[name]['traffic_over_algorithmic']:.2f}")
    out.update(stamp(MEMBOUND_SOURCES))  # tests/test_profiles_fresh.py: a profile is quoted only for the device sources it was measured on
    (OUT / f"{R}_membound_rocprof.json").write_text(json.dumps(out, indent=1))
    stats = glob.glob(str(OUT / "mbT" / "**" / "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], OUT / f"{R}_membound_kernel_stats.csv")
        (OUT / f"{R}_membound_kernel_stats.stamp.json").write_text(json.dumps(stamp(MEMBOUND_SOURCES), indent=1))


if __name__ == "__main__":
    main()
