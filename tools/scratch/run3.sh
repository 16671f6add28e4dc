set -uo pipefail
cd "${GRAFT_REPO_ROOT:?}"
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > gpurun_out/r06_gputest2.log; tail -12 gpurun_out/r06_gputest2.log
