set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r06_gputest1.log; tail -3 gpurun_out/r06_gputest1.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/bench_r06_a.json 2> gpurun_out/bench_r06_a.err; tail -c 400 gpurun_out/bench_r06_a.err
python tools/vendor_yardstick.py --json gpurun_out/r06_vendor_yardstick.json > /dev/null 2> gpurun_out/r06_vendor_yardstick.err; tail -8 gpurun_out/r06_vendor_yardstick.err
