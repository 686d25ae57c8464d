#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3head_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r3head_suite.log
for S in 1 8 32; do bash tools/runs/r3_small.sh r3head_s$S $S 3 AEV_FUSED_HEAD=0 AEV_FUSED_HEAD=1 | sed "s/^/S=$S /"; done
