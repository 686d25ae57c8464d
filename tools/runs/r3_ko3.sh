#!/bin/bash
# knock-out library x work queue on / off on conv2: is the queue's atomic counter the floor?
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/ko3; mkdir -p $O
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in ab/ko14.so ab/bias.so ab/carry.so; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  for E in "AEV_X=1" "AEV_NO_QUEUE=1"; do
    env $E AEV_NO_OVERLAP=1 AEV_STAMPS=3 python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "AEV_STAMPS.*blocks (" | tail -1 | cut -c1-200 | sed "s|^|$v $E |"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
