#!/bin/bash
# per-kernel times of the long-segment radix path under library variants: ab_sort.sh NSEG N lib ... ("" = the product); rocprofv3 kernel stats per variant
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
NSEG=$1; N=$2; shift 2; KIND=${KIND:-f32}
for lib in "$@"; do
  name=$(basename "${lib:-product}" .so)
  out=gpurun_out/ab_sort/${NSEG}x${N}_${KIND}_$name
  rm -rf $out; mkdir -p $out
  KF_HIP_LIB=${lib:+$PWD/$lib} timeout 150 rocprofv3 --kernel-trace --stats -d $out -o s -- python3 tools/scratch/sort_case.py $NSEG $N 5 $KIND > $out/log.txt 2>&1
  echo "== $name ($NSEG x $N)"
  python3 tools/scratch/rocpd_stats.py $out/s_results.db
done
