#!/bin/bash
# un-profiled: when does a tile-kernel layer start relative to the end of the k_post it waits for? (one stream, last step of a run)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
for pair in "1 3" "3 5" "5 7" "0 3"; do
  set -- $pair
  AEV_STAMPS_ABS=1 AEV_POST_STAMPS=$1 AEV_STAMPS=$2 python bench.py --streams 1 --steps 40 --warmup 10 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "absolute" | tail -2 | sed "s|^|post $1 tile $2: |"
done
