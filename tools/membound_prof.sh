# rocprofv3 evidence for the memory-bound half of the path: per-kernel durations from a --kernel-trace --stats
# pass and bytes beyond L2 from separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (MI355X_MICROARCH.md, HBM section) over
# tools/membound_bench.py --markers; tools/membound_prof.py cuts the dispatch lists at the markers and writes
# profiles/$R_membound_rocprof.json + profiles/$R_membound_kernel_stats.csv. The program stands directly behind `--`.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r04}"
rm -rf gpurun_out/mbT gpurun_out/mbF gpurun_out/mbW
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/mbT -o r -- python3 tools/membound_bench.py --markers --rounds 5 --json gpurun_out/${R}_membound.json > gpurun_out/mbT.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/mbF -o r -- python3 tools/membound_bench.py --markers --rounds 5 > gpurun_out/mbF.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/mbW -o r -- python3 tools/membound_bench.py --markers --rounds 5 > gpurun_out/mbW.log 2>&1
python3 tools/membound_prof.py "$R"
