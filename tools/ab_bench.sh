#!/bin/bash
# A/B of two builds of libaev_hip.so on ONE box, interleaved (run-to-run and box-to-box noise is ~1.5 %):
#   tools/ab_bench.sh ab/old.so ab/new.so [rounds]
A=$1; B=$2; R=${3:-3}
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for r in $(seq $R); do
  for lib in $A $B; do
    cp $lib async_ev_cnn_amd/libaev_hip.so
    v=$(python bench.py --no-cpu-baseline --no-single --no-host-feed --profile-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$lib $v"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
