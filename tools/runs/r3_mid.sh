#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
for S in 160 192 256; do bash tools/runs/r3_small.sh r3mid_s$S $S 2 - AEV_OVERLAP=0+AEV_HOIST_MAX=1000 AEV_OVERLAP=0+AEV_HOIST_MAX=1000+AEV_LOOKAHEAD=0 | sed "s/^/S=$S /"; done
