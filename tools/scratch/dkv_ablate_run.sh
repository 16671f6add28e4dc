#!/bin/bash
cd "$(dirname "$0")/../.."
for lib in "" $(ls tools/scratch/lib_dkv_*.so); do
  printf "%-46s" "${lib:-product}"
  KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --rounds 5 2>&1 | grep attn_bwd_dkv
  printf "%-46s" "   zeros"
  KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --rounds 5 --zeros 2>&1 | grep attn_bwd_dkv
done
echo "S sweep (product)"
python tools/attn_bench.py --rounds 3 --B 4 --S 8192 2>&1 | grep attn_bwd_dkv
python tools/attn_bench.py --rounds 3 --B 2 --S 16384 2>&1 | grep attn_bwd_dkv
python tools/attn_bench.py --rounds 3 --B 16 --S 2048 2>&1 | grep attn_bwd_dkv
