#!/bin/bash
# SQ counters of the radix kernels for one variant: pmc_sort.sh lib NSEG N
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
lib=$1; NSEG=$2; N=$3
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  out=gpurun_out/pmc_sort/p$i
  rm -rf $out; mkdir -p $out
  KF_HIP_LIB=${lib:+$PWD/$lib} timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $out -o r -- python3 tools/scratch/sort_case.py $NSEG $N 2 > $out/log.txt 2>&1
  f=$(find $out -name "r_counter_collection.csv" | head -1)
  python3 - "$f" <<'P'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
for k, v in acc.items():
    print(k, "dispatches", len(n[k]), {c: round(x / len(n[k])) for c, x in v.items()})
P
done
