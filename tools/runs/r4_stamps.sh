#!/bin/bash
# In-kernel stamps (AEV_STAMPS) of the tile-kernel layers (3 5 7 = conv2 .. conv4 of the e-fcn), with optional env settings.
#   usage: tools/runs/r4_stamps.sh <tag> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; shift; O=gpurun_out/$T; mkdir -p $O
for L in 3 5 7; do env "$@" AEV_STAMPS=$L python bench.py --steps 6 --warmup 2 --profile-steps 0 --sustained-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy > /dev/null 2> $O/stamps_$L.txt; echo "layer $L"; grep AEV_STAMPS $O/stamps_$L.txt | tail -4; done
