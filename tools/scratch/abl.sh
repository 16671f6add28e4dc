#!/bin/bash
set -uo pipefail
for i in 1 2; do
for v in "" base nosm nobar nodma; do
  if [ -z "$v" ]; then echo "== new"; timeout 300 python tools/attn_bench.py --rounds 4 --no-bwd ${AB_ARGS:-} 2>&1 | grep -E "attn_fwd"
  else echo "== $v"; KF_HIP_LIB=$PWD/tools/scratch/lib_$v.so timeout 300 python tools/attn_bench.py --rounds 4 --no-bwd ${AB_ARGS:-} 2>&1 | grep -E "attn_fwd"; fi
done; done
