#!/bin/bash
# BASELINE config 3 (dense 1 Mevent/s, B = 200) A/B of environment settings / libraries
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
N=$1; shift
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for r in $(seq $N); do
  for v in "$@"; do
    lib=""; envs=""; IFS='+' read -ra parts <<< "$v"
    for p in "${parts[@]}"; do if [[ "$p" == *.so ]]; then lib=$p; elif [[ "$p" != "-" ]]; then envs="$envs $p"; fi; done
    [[ -n "$lib" ]] && cp $lib async_ev_cnn_amd/libaev_hip.so || cp $keep async_ev_cnn_amd/libaev_hip.so
    ms=$(env $envs python bench.py --kind dense1M --batch 200 --no-cpu-baseline --no-single --no-host-feed --no-dense --profile-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "cfg3 $v $ms"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
