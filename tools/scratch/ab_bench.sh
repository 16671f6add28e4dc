#!/bin/bash
# same-box A/B of library variants under bench.py itself (the whole step): ab_bench.sh ROUNDS lib1.so lib2.so ... ("" = the product library)
cd "$(dirname "$0")/../.."
R=$1; shift
for r in $(seq $R); do
  for lib in "$@"; do
    KF_ALLOW_STALE_LIB=1 KF_HIP_LIB=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline --no-ceiling --sustain-seconds 0 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-32s' % '${lib:-product}', round(j['ms_per_step'],3), ' '.join('%s %.3f' % (k.replace('attn_','').replace('_mfma',''), v['avg_ms']) for k,v in j['kernels'].items()))"
  done
done
