#!/bin/bash
# A/B of two builds at small stream counts (latency mode): tools/ab_small.sh "1 4 16" ab/old.so ab/new.so [ENV=val]
SS=$1; A=$2; B=$3; shift 3
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for r in 1 2; do for S in $SS; do for v in $A $B; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  ms=$(env "$@" python bench.py --streams $S --steps 200 --warmup 20 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "S=$S $v $ms"
done; done; done
cp $keep async_ev_cnn_amd/libaev_hip.so
