#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider -x > gpurun_out/r3few_suite.log 2>&1; echo "suite rc=$?"; tail -4 gpurun_out/r3few_suite.log
bash tools/runs/r3_small.sh r3few_s1 1 2 AEV_FEW_TILE_CAP=128+AEV_NO_M16=1 AEV_FEW_TILE_CAP=128 AEV_FEW_TILE_CAP=64 AEV_FEW_TILE_CAP=32 AEV_FEW_TILE_CAP=48 | sed "s/^/S=1 /"
for S in 8 32 64; do bash tools/runs/r3_small.sh r3few_s$S $S 2 AEV_FEW_TILE_CAP=128+AEV_NO_M16=1 AEV_FEW_TILE_CAP=64 | sed "s/^/S=$S /"; done
