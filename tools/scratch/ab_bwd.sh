#!/bin/bash
# same-box A/B of library variants on the C3 attention step: ab_bwd.sh ROUNDS lib1.so lib2.so ... ("" = the product library); interleaved, one process per run
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    printf "%-34s" "${lib:-product}"
    KF_ALLOW_STALE_LIB=1 KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --rounds 6 2>&1 | grep "attn_" | awk '{printf "%s %s  ", $1, $3}'; echo
  done
done
