#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
for S in 4 16 31; do bash tools/runs/r3_small.sh r3postlds_s$S $S 2 - AEV_POST_LDS_MIN=0 | sed "s/^/S=$S /"; done
