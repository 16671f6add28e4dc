# Matrix-pipe utilisation and wave-cycle breakdown of the bench kernels: rocprofv3 --pmc passes (SQ counters only, no other
# trace domains) over `bench.py --steps 3 --warmup 1`, aggregated into profiles/r01_pmc_util.json by tools/pmc_util.py
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r04}"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf gpurun_out/pu$i
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pu$i -o r -- python3 bench.py --steps 3 --warmup 1 --sustain-seconds 0 --no-cpu-baseline --no-ceiling > gpurun_out/pu$i.log 2>&1
done
python3 tools/pmc_util.py "$R"
