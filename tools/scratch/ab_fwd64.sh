#!/bin/bash
# same-box A/B of forward variants at head size 64: ab_fwd64.sh ROUNDS lib ... ("" = the product)
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    printf "%-36s" "${lib:-product}"
    KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --B 8 --H 64 --D 64 --no-bwd --rounds 5 2>&1 | grep "attn_fwd" | tr '\n' ' '; echo
  done
done
