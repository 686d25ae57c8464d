#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3hoist4_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r3hoist4_suite.log
for S in 1 16 48; do bash tools/runs/r3_small.sh r3hoist4_s$S $S 2 AEV_TAIL_SMALL=0 AEV_TAIL_SMALL=1 | sed "s/^/S=$S /"; done
