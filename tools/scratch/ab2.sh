#!/bin/bash
set -uo pipefail
timeout 600 python -m pytest tests/test_gpu_attention.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
for z in "--zeros" ""; do for i in 1 2; do
  echo "== base $z"; KF_HIP_LIB=$PWD/tools/scratch/lib_base.so timeout 300 python tools/attn_bench.py --rounds 4 --no-bwd $z 2>&1 | grep -E "attn_fwd"
  echo "== new $z";  timeout 300 python tools/attn_bench.py --rounds 4 --no-bwd $z 2>&1 | grep -E "attn_fwd"
done; done
timeout 200 python tools/attn_timeline.py 2>&1 | tail -11
