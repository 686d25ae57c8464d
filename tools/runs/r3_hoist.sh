#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r3hoist; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
for S in 1 4 16 32; do bash tools/runs/r3_small.sh r3hoist_s$S $S 2 AEV_HOIST_MAX=0 AEV_HOIST_MAX=64 | sed "s/^/S=$S /"; done
