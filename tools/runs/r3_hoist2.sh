#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
for S in 48 63; do bash tools/runs/r3_small.sh r3hoist_s$S $S 2 AEV_HOIST_MAX=0 AEV_HOIST_MAX=64 | sed "s/^/S=$S /"; done
bash tools/runs/r3_timeline.sh r3tl1h 1 AEV_HOIST_MAX=64 | head -40
