#!/bin/bash
# same-box A/B of two device-library builds: tools/scratch/lib_base.so (KF_HIP_LIB) against the in-tree library, alternating
set -uo pipefail
ARGS="${AB_ARGS:-}"
for i in 1 2 3; do
  echo "== base"; KF_HIP_LIB=$PWD/tools/scratch/lib_base.so timeout 300 python tools/attn_bench.py --rounds 4 $ARGS 2>&1 | grep -E "attn_(fwd|bwd)" | head -6
  echo "== new";  timeout 300 python tools/attn_bench.py --rounds 4 $ARGS 2>&1 | grep -E "attn_(fwd|bwd)" | head -6
done
