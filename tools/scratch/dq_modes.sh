#!/bin/bash
cd "$(dirname "$0")/../.."
for r in 1 2 3; do for m in ${MODES:-plain arena}; do MODE=$m timeout 200 python tools/scratch/dq_modes.py 2>&1 | tail -1; done; done
