#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/la1; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "kernel or efcn or tail or lookahead or config3 or many_streams or random_arch" > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 $O/parity.log
[ $rc -ne 0 ] && exit 1
AB_KERNEL=k_conv_patch tools/ab_bench.sh 3 AEV_NO_LA1=1 - | tee $O/ab.txt
for E in AEV_NO_LA1=1 AEV_X=1; do
  for L in 5 7 9; do env $E AEV_NO_OVERLAP=1 AEV_STAMPS=$L python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "AEV_STAMPS.*blocks (" | tail -1 | cut -c1-200 | sed "s|^|$E L=$L |"; done
done
