#!/bin/bash
# on the GPU box: the forward's time per ablation variant built by fwd_w4_ablate.sh (same box, one process each)
cd "$(dirname "$0")/../.."
for lib in "" $(ls tools/scratch/lib_w4_*.so); do
  echo "=== ${lib:-full}"
  KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --no-bwd --rounds 5 2>&1 | grep attn_fwd
  KF_HIP_LIB=${lib:+$PWD/$lib} python tools/attn_bench.py --no-bwd --rounds 5 --zeros 2>&1 | grep attn_fwd | sed 's/^/   zeros/'
done
echo "=== full, S sweep"
python tools/attn_bench.py --no-bwd --rounds 3 --B 4 --S 8192 2>&1 | grep attn_fwd
python tools/attn_bench.py --no-bwd --rounds 3 --B 2 --S 16384 2>&1 | grep attn_fwd
python tools/attn_bench.py --no-bwd --rounds 3 --B 16 --S 2048 2>&1 | grep attn_fwd
echo "=== v3"
KF_ATTN_FWD_V3=1 python tools/attn_bench.py --no-bwd --rounds 5 --variants KF_ATTN_FWD_V3 2>&1 | grep attn_fwd
