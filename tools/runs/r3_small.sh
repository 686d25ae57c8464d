#!/bin/bash
# single-stream (latency mode) A/B of environment settings: r3_small.sh <tag> <streams> <rounds> <variant> ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; S=$2; N=$3; shift; shift; shift
O=gpurun_out/$T; mkdir -p $O
for r in $(seq $N); do
  for v in "$@"; do
    envs=""; IFS='+' read -ra parts <<< "$v"; for p in "${parts[@]}"; do [[ "$p" != "-" ]] && envs="$envs $p"; done
    ms=$(env $envs python bench.py --streams $S --steps 200 --warmup 20 --no-cpu-baseline --no-single --no-host-feed --no-dense --profile-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$v $ms" | tee -a $O/ab.log
  done
done
