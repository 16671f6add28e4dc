cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python bench.py > gpurun_out/bench_v6.json 2> gpurun_out/bench_v6.err; tail -c 600 gpurun_out/bench_v6.err | grep -v NCCL | tail -3
rm -rf gpurun_out/prof_v6
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v6 -o r01 -- python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_v6_prof.json 2> gpurun_out/prof_v6.err
find gpurun_out/prof_v6 -name "*kernel_stats.csv" | head
