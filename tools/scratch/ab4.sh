#!/bin/bash
set -uo pipefail
for i in 1 2; do
  echo "== new --zeros";  timeout 300 python tools/attn_bench.py --rounds 4 --zeros 2>&1 | grep -E "dkv"
  echo "== novalu --zeros"; KF_HIP_LIB=$PWD/tools/scratch/lib_novalu.so timeout 300 python tools/attn_bench.py --rounds 4 --zeros 2>&1 | grep -E "dkv"
done
