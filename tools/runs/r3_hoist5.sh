#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/r3hoist5_suite.log 2>&1; echo "suite rc=$?"; tail -6 gpurun_out/r3hoist5_suite.log
bash tools/runs/r3_small.sh r3hoist5_s1 1 1 - 
bash tools/runs/r3_small.sh r3hoist5_s256 256 1 -
