#!/bin/bash
# few-streams A/B of libraries (+ env): r3_small_lib.sh <streams> <rounds> lib[+ENV=..] ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
S=$1; N=$2; shift 2
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for r in $(seq $N); do
  for v in "$@"; do
    lib=""; envs=""; IFS='+' read -ra parts <<< "$v"
    for p in "${parts[@]}"; do if [[ "$p" == *.so ]]; then lib=$p; elif [[ "$p" != "-" ]]; then envs="$envs $p"; fi; done
    [[ -n "$lib" ]] && cp $lib async_ev_cnn_amd/libaev_hip.so || cp $keep async_ev_cnn_amd/libaev_hip.so
    ms=$(env $envs python bench.py --streams $S --steps 200 --warmup 20 --no-cpu-baseline --no-single --no-host-feed --no-dense --profile-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "S=$S $v $ms"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
