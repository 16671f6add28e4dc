# matrix-pipe utilisation, VALU / MFMA, clock and traffic beyond L2 of the GEMM kernels at BASELINE configs C2 / C4 and
# around them: rocprofv3 --pmc passes (SQ / TCC counters only, no other trace domain), ONE process per case and pass with the program
# directly behind `--`; aggregated into profiles/$R_gemm_pmc.json by tools/gemm_pmc.py
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r06}"
rm -rf gpurun_out/gp
mkdir -p gpurun_out/gp
i=0
python3 - <<'PY' > gpurun_out/gp/cases.txt
import sys
sys.path.insert(0, "tools")
import gemm_pmc_case
print("\n".join(gemm_pmc_case.CASES))
PY
while IFS= read -r CASE; do
  i=$((i+1))
  p=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_INSTS_VALU GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
    p=$((p+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/gp/c${i}p${p} -o r -- python3 tools/gemm_pmc_case.py "$CASE" > gpurun_out/gp/c${i}p${p}.log 2>&1 || echo "case $i pass $p failed" >&2
  done
  echo "$i|$CASE" >> gpurun_out/gp/index.txt
done < gpurun_out/gp/cases.txt
python3 tools/gemm_pmc.py "$R"
