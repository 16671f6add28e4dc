set -euo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r06}"
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python bench.py > gpurun_out/bench_$R.json 2> gpurun_out/bench_$R.err || true; tail -c 600 gpurun_out/bench_$R.err | grep -v NCCL | tail -3 || true
rm -rf gpurun_out/prof_$R
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o $R -- python3 bench.py --steps 10 --warmup 3 --sustain-seconds 0 --no-cpu-baseline --no-ceiling > gpurun_out/bench_${R}_prof.json 2> gpurun_out/prof_$R.err
S=$(find gpurun_out/prof_$R -name "*kernel_stats.csv" | head -1)
cp "$S" gpurun_out/${R}_bench_kernel_stats.csv
python3 -c "import json, bench; print(json.dumps(bench.stamp(bench.BENCH_SOURCES), indent=1))" > gpurun_out/${R}_bench_kernel_stats.stamp.json   # sidecar: which device sources the CSV was measured on
echo "$S"
