#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
for L in 0 1 3 5 7 9; do AEV_POST_STAMPS=$L python bench.py --streams ${1:-1} --steps 30 --warmup 10 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep POST_STAMPS | tail -1; done
