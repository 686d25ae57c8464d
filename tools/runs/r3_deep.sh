#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3deep_suite.log 2>&1; echo "suite rc=$?"; tail -5 gpurun_out/r3deep_suite.log
for S in 1 8 32 96; do bash tools/runs/r3_small.sh r3deep_s$S $S 2 AEV_TAIL_FROM_FEW=0 - AEV_TAIL_FROM_FEW=7 | sed "s/^/S=$S /"; done
