#!/bin/bash
# eager per-launch microseconds of the bookkeeping kernel, layer by layer, for a list of library builds / switches:  r5_post_eager.sh <variant> ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in "$@"; do
  lib=""; envs=""
  IFS='+' read -ra parts <<< "$v"
  for p in "${parts[@]}"; do if [[ "$p" == *.so ]]; then lib=$p; elif [[ "$p" != "-" ]]; then envs="$envs $p"; fi; done
  if [[ -n "$lib" ]]; then cp $lib async_ev_cnn_amd/libaev_hip.so; else cp $keep async_ev_cnn_amd/libaev_hip.so; fi
  for S in ${STREAMS:-256 1}; do
  env $envs python bench.py --streams $S --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 0 --steps 10 --profile-steps 6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
pl=d['us_per_launch_by_layer']
print('$v', 'S=$S', 'ms/step', d['ms_per_step'], ' '.join('%s %.1f' % (k, v['k_compact']) for k, v in pl.items() if 'k_compact' in v), ' pools:', ' '.join('%s %.1f' % (k, v['k_pool_eval']) for k, v in pl.items() if 'k_pool_eval' in v))"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
