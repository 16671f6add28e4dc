# HBM-side traffic of the bench kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (MI355X_MICROARCH.md),
# aggregated per kernel into profiles/$R_pmc_traffic.json by tools/pmc_traffic.py
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r04}"
rm -rf gpurun_out/pmcF gpurun_out/pmcW
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcF -o r -- python3 bench.py --steps 3 --warmup 1 --sustain-seconds 0 --no-cpu-baseline --no-ceiling > gpurun_out/pmcF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmcW -o r -- python3 bench.py --steps 3 --warmup 1 --sustain-seconds 0 --no-cpu-baseline --no-ceiling > gpurun_out/pmcW.log 2>&1
python3 tools/pmc_traffic.py "$R"
