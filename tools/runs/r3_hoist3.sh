#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3hoist3_suite.log 2>&1; echo "suite rc=$?"; tail -2 gpurun_out/r3hoist3_suite.log
for S in 1 16; do bash tools/runs/r3_small.sh r3hoist3_s$S $S 2 AEV_HOIST_EDGES=9 AEV_HOIST_EDGES=0 AEV_HOIST_EDGES=1 AEV_HOIST_EDGES=2 AEV_HOIST_EDGES=3 | sed "s/^/S=$S /"; done
