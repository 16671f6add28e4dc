cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" 2>&1 | tail -1
python bench.py > gpurun_out/bench_v10.json 2> gpurun_out/bench_v10.err; tail -c 600 gpurun_out/bench_v10.err | grep -v NCCL | tail -3
rm -rf gpurun_out/prof_v10
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v10 -o r01 -- python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_v10_prof.json 2> gpurun_out/prof_v10.err
find gpurun_out/prof_v10 -name "*kernel_stats.csv" | head
