#!/bin/bash
# same-box A/B of forward library variants: ab_libs.sh ROUNDS lib1.so lib2.so ... ("" = the product library); interleaved, one process per run
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    printf "%-40s" "${lib:-product}"
    KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --no-bwd --rounds 5 2>&1 | grep attn_fwd_mfma
  done
done
