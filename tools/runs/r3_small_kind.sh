#!/bin/bash
# few-streams A/B on another stream kind: r3_small_kind.sh <kind> <batch> <streams> <rounds> variants...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
K=$1; B=$2; S=$3; N=$4; shift 4
for r in $(seq $N); do
  for v in "$@"; do
    envs=""; IFS='+' read -ra parts <<< "$v"; for p in "${parts[@]}"; do [[ "$p" != "-" ]] && envs="$envs $p"; done
    ms=$(env $envs python bench.py --kind $K --batch $B --streams $S --steps 200 --warmup 20 --no-cpu-baseline --no-single --no-host-feed --no-dense --profile-steps 0 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$K B=$B S=$S $v $ms"
  done
done
