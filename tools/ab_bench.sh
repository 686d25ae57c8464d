#!/bin/bash
# Interleaved A/B/... of bench.py variants on ONE box (run-to-run and box-to-box noise is ~1.5 %).  A variant is a build of
# libaev_hip.so ("ab/old.so") or an environment setting ("AEV_OVERLAP=0", "-" for none), optionally both ("ab/x.so+AEV_OVERLAP=0"):
#   tools/ab_bench.sh 3 ab/old.so ab/new.so          tools/ab_bench.sh 3 - AEV_OVERLAP=0
#   AB_KERNEL=k_leak_sweep tools/ab_bench.sh 2 ab/a.so ab/b.so    (adds that kernel class's us per step)
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
    if [[ -n "$AB_KERNEL" ]]; then # also the per-step time of one kernel class (live per-kernel timing, 6 eager steps)
      ms=$(env $envs python bench.py --no-cpu-baseline --no-single --no-host-feed --no-dense --sustained-steps ${AB_SUSTAINED:-200} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['sustained']['ms_per_step'], round(d['kernels']['$AB_KERNEL']['ms_total']*1e3/6, 1))")
    else
      ms=$(env $envs python bench.py --no-cpu-baseline --no-single --no-host-feed --no-dense --profile-steps 0 --sustained-steps ${AB_SUSTAINED:-200} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['sustained']['ms_per_step'])")
    fi
    echo "$v $ms"
  done
done
cp $keep async_ev_cnn_amd/libaev_hip.so
