set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r02}"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INST_CYCLES_VMEM_RD" "SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf gpurun_out/pk$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pk$i -o r -- python3 tools/attn_bench.py --rounds 2 --variants default > gpurun_out/pk$i.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
for i in range(1,5):
    f=glob.glob(f'gpurun_out/pk{i}/*counter_collection.csv')
    if not f: print('no file',i); continue
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f[0])):
        k=row['Kernel_Name']
        if 'dkv' not in k: continue
        acc[k[:40]][row['Counter_Name']].append(float(row['Counter_Value']))
    for k,d in acc.items():
        print(k, {c: sum(v)/len(v) for c,v in d.items()})
PY
