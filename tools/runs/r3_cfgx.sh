#!/bin/bash
# Round 3: a tile-kernel instance on one layer by environment.  usage: r3_cfgx.sh <tag> <rounds> "ENV=.. ENV=.." ...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; N=$2; shift 2; O=gpurun_out/$T; mkdir -p $O
for v in "$@"; do
  [ "$v" = "-" ] && continue
  env ${v//+/ } python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "efcn or config3 or tail or lookahead" > $O/parity.log 2>&1; rc=$?; echo "parity [$v] rc=$rc"; tail -2 $O/parity.log
  [ $rc -ne 0 ] && exit 1
done
AB_KERNEL=k_conv_patch tools/ab_bench.sh $N "$@" | tee $O/ab.txt
for v in "$@"; do
  e=""; [ "$v" != "-" ] && e=${v//+/ }
  for L in 3; do env $e AEV_NO_OVERLAP=1 AEV_STAMPS=$L python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "AEV_STAMPS" | grep -v "percentile\|CUs held" | tail -2 | cut -c1-330; done
done
