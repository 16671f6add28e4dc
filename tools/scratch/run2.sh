set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_attention.py -x -q -k "ragged or mfma_path or generated_dkv or key_blocks or bounded or without_an_lse" 2>&1 | tail -25 > gpurun_out/r06_ragged1.log; tail -25 gpurun_out/r06_ragged1.log
