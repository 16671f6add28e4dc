#!/bin/bash
# A/B of the CU-partitioned backward (KF_ATTN_BWD_SPLIT_CUS / _GROUPS) inside bench.py's step, interleaved on one box.
cd /root/repo
OUT=gpurun_out/ab_split.txt
: > $OUT
for rep in 1 2; do
for cfg in "0 32 32" "64 24 64" "64 48 64" "96 40 96" "128 32 32" "64 96 64" "96 80 96" "64 24 160"; do
  set -- $cfg
  if [ "$1" = "0" ]; then unset KF_ATTN_BWD_SPLIT_CUS; else export KF_ATTN_BWD_SPLIT_CUS=$1; fi
  export KF_ATTN_BWD_SPLIT_GROUPS=$2 KF_ATTN_BWD_SPLIT_P0=$3
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-ceiling --sustain-seconds 1 > gpurun_out/_b.json 2> gpurun_out/_b.err
  python - "$cfg" >> $OUT <<'PY'
import json,sys
try:
    d=json.load(open('gpurun_out/_b.json'))
    k=d['kernels']
    print(sys.argv[1], 'ms/step %.3f sus %.3f prof %.3f'%(d['ms_per_step'],d.get('ms_per_step_sustained',0),d['ms_per_step_profiled']),
          ' '.join('%s=%.3fx%d'%(n.replace('attn_',''),v['avg_ms'],v['launches']//30) for n,v in k.items() if 'attn' in n), 'checks', all(d['checks'].values()))
except Exception as e:
    print(sys.argv[1], 'FAILED', e, open('gpurun_out/_b.err').read()[-600:])
PY
done
done
cat $OUT
