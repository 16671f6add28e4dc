# Everything DESIGN.md quotes for a round, collected in ONE gpurun call (about 25 GPU-minutes): tests + smoke + bench + rocprof stats
# (tools/roundend.sh), PMC traffic / utilisation of the bench kernels, the memory-bound table, the GEMM sweep with its counter passes,
# attention at head size 64, large logits, parity margins, config C5's forms, the two timelines; then bench.py once more so that the
# line quotes the traffic profile just taken. Outputs land in gpurun_out/ (merged back); copy what is to be judged into profiles/.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:?GRAFT_REPO_ROOT is not set (run under gpurun)}"
export TMPDIR=/tmp
R="${KF_ROUND:-r06}"
export KF_ROUND=$R
bash tools/roundend.sh > gpurun_out/${R}_roundend.log 2>&1; tail -5 gpurun_out/${R}_roundend.log
bash tools/pmc_traffic.sh > gpurun_out/${R}_pmc_traffic.log 2>&1; cp profiles/${R}_pmc_traffic.json gpurun_out/ 2>/dev/null
bash tools/pmc_util.sh > gpurun_out/${R}_pmc_util.log 2>&1; cp profiles/${R}_pmc_util.json gpurun_out/ 2>/dev/null
bash tools/membound_prof.sh > gpurun_out/${R}_membound.log 2>&1   # (writes its results into gpurun_out/ itself)
bash tools/gemm_pmc.sh > gpurun_out/${R}_gemm_pmc.log 2>&1; cp profiles/${R}_gemm_pmc.json gpurun_out/ 2>/dev/null
python tools/gemm_sweep.py --json gpurun_out/${R}_gemm_sweep.json > gpurun_out/${R}_gemm_sweep.txt 2>&1
python tools/attn_large_logits.py > gpurun_out/${R}_attn_large_logits.txt 2>&1
python tools/attn_large_logits.py --D 64 --forms default,v3v4 > gpurun_out/${R}_attn_large_logits_d64.txt 2>&1
python tools/attn_bench.py --B 8 --H 64 --D 64 --rounds 5 --variants default,KF_ATTN_FWD_V3=1+KF_ATTN_DKV_V4=1 > gpurun_out/${R}_attn_d64.txt 2>&1
python tools/attn_parity_margins.py --out gpurun_out/${R}_attn_parity_margins.json > gpurun_out/${R}_attn_parity_margins.log 2>&1
for F in reference fused fused-norm; do python tools/block_bench.py --form $F --steps 20 --json gpurun_out/${R}_block_c5_$F.json > /dev/null 2> gpurun_out/${R}_block_$F.err; done
python tools/block_bench.py --form fused --force-comm --check --steps 20 --json gpurun_out/${R}_block_c5_check_1gpu.json > /dev/null 2> gpurun_out/${R}_block_check.err
python tools/attn_dkv_w4_timeline.py > gpurun_out/${R}_attn_dkv_timeline.txt 2>&1
python tools/attn_fwd_w4_timeline.py > gpurun_out/${R}_attn_fwd_timeline.txt 2>&1
python tools/attn_ragged_bench.py --json gpurun_out/${R}_attn_ragged.json > gpurun_out/${R}_attn_ragged.txt 2>&1
python tools/attn_ragged_bench.py --D 64 --H 64 --S 4096,4000,1000 >> gpurun_out/${R}_attn_ragged.txt 2>&1
python tools/vendor_yardstick.py --json gpurun_out/${R}_vendor_yardstick.json > /dev/null 2> gpurun_out/${R}_vendor_yardstick.err
cp gpurun_out/${R}_pmc_traffic.json profiles/ 2>/dev/null
python bench.py > gpurun_out/bench_${R}_final.json 2> gpurun_out/bench_${R}_final.err
tail -c 900 gpurun_out/bench_${R}_final.json
