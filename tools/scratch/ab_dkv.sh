#!/bin/bash
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    printf "%-40s" "${lib:-product}"
    KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --rounds 5 2>&1 | grep "attn_bwd_dkv\|attn_bwd_dq" | tr '\n' ' '; echo
  done
done
