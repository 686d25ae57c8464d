#!/bin/bash
# knock-out library x blocks per CU on one layer: is the skeleton latency- or issue-bound?
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/ko2; mkdir -p $O
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in ab/ko14.so ab/ko2.so ab/bias.so; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  for B in 1 2 3; do
    AEV_BLOCKS_3=$B AEV_NO_OVERLAP=1 AEV_STAMPS=3 python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "AEV_STAMPS.*blocks (" | tail -1 | cut -c1-200 | sed "s|^|$v B=$B |"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
