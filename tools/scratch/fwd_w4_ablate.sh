#!/bin/bash
# Timing variants of the generated forward (tools/gen_attn_fwd.py --ablate ...): one library per variant under tools/scratch/, for
# same-box A/B through KF_HIP_LIB (tools/attn_bench.py --no-bwd). Results of the ablated builds are WRONG by construction.
set -e
cd "$(dirname "$0")/../.."
for v in "$@"; do
  name=$(echo "$v" | tr ',' '_')
  python tools/gen_attn_fwd.py --ablate "$v" --out "$PWD/kfunca_amd/_build/attn_fwd_w4_$name.inc" > /dev/null
  python tools/scratch/build_variant.py "w4_$name" attention.hip "-DKF_FWD_W4_INC=\"$PWD/kfunca_amd/_build/attn_fwd_w4_$name.inc\"" &
done
wait
ls -la tools/scratch/lib_w4_*.so
