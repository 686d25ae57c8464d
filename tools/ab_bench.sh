#!/bin/bash
# Interleaved A/B/... of bench.py variants on ONE box (run-to-run and box-to-box noise is ~1.5 %).  A variant is a build of
# libaev_hip.so ("ab/old.so") or an environment setting ("AEV_OVERLAP=0", "-" for none), optionally both ("ab/x.so+AEV_OVERLAP=0"):
#   tools/ab_bench.sh 3 ab/old.so ab/new.so          tools/ab_bench.sh 3 - AEV_OVERLAP=0
# Prints per run: headline ms per step, sustained ms per step (AB_SUSTAINED steps, default 200), and -- with AB_LAYERS=1 -- the eager
# per-launch microseconds of the conv layers' kernels (bench.py's per-kernel pass).
R=$1; shift
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for r in $(seq $R); do
  for v in "$@"; do
    lib=""; envs=""
    IFS='+' read -ra parts <<< "$v"
    for p in "${parts[@]}"; do
      if [[ "$p" == *.so ]]; then lib=$p; elif [[ "$p" != "-" ]]; then envs="$envs $p"; fi
    done
    if [[ -n "$lib" ]]; then cp $lib async_ev_cnn_amd/libaev_hip.so; else cp $keep async_ev_cnn_amd/libaev_hip.so; fi
    ps=0; [[ -n "$AB_LAYERS" ]] && ps=6
    ms=$(env $envs python bench.py --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --profile-steps $ps --sustained-steps ${AB_SUSTAINED:-200} ${AB_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
out=[d['ms_per_step'], d['sustained']['ms_per_step'] if d.get('sustained') else None]
bl=(d.get('roofline') or {}).get('by_layer') or {}
out += ['%s %.1f' % (k, v['us_per_launch']) for k, v in bl.items()]
print(*out)")
    echo "$v $ms"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
