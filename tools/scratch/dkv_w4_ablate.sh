#!/bin/bash
# Timing variants of the generated dK/dV (tools/gen_attn_dkv.py --ablate ...): one library per variant under tools/scratch/, for
# same-box A/B through KF_HIP_LIB (tools/attn_bench.py). Results of the ablated builds are WRONG by construction.
set -e
cd "$(dirname "$0")/../.."
for v in "$@"; do
  name=$(echo "$v" | tr ',' '_')
  python tools/gen_attn_dkv.py --ablate "$v" --out "$PWD/kfunca_amd/_build/attn_dkv_w4_$name.inc" > /dev/null 2>&1
  python tools/scratch/build_variant.py "dkv_$name" attention.hip "-DKF_DKV_W4_INC=\"$PWD/kfunca_amd/_build/attn_dkv_w4_$name.inc\"" > /dev/null &
done
wait
ls tools/scratch/lib_dkv_*.so
