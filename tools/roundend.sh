cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
true
python bench.py > gpurun_out/bench_v4.json 2> gpurun_out/bench_v4.err; tail -c 600 gpurun_out/bench_v4.err | grep -v NCCL | tail -3
rm -rf gpurun_out/prof_v4
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_v4 -o r01 -- python3 bench.py --steps 10 --warmup 3 > gpurun_out/bench_v4_prof.json 2> gpurun_out/prof_v4.err
find gpurun_out/prof_v4 -name "*kernel_stats.csv" | head
