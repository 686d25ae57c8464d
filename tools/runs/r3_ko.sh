#!/bin/bash
# Round 3: knock-out builds of the tile kernel (AEV_KO, timing only).  usage: r3_ko.sh <tag> <layer> libs...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; L=$2; shift 2; O=gpurun_out/$T; mkdir -p $O
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in "$@"; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  AEV_NO_OVERLAP=1 AEV_STAMPS=$L python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > /dev/null 2> $O/stamps_$(basename $v .so).txt
  echo "== $v"; grep AEV_STAMPS $O/stamps_$(basename $v .so).txt | grep -v "percentile\|CUs held" | tail -2
done
cp $keep async_ev_cnn_amd/libaev_hip.so
