#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/cfg5; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "kernel or efcn" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -3 $O/parity.log
AEV_CFG_3=5 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "efcn or config3" > $O/parity5.log 2>&1; echo "parity cfg5 rc=$?"; tail -3 $O/parity5.log
AB_KERNEL=k_conv_patch tools/ab_bench.sh 2 - AEV_CFG_3=5 AEV_CFG_3=5+AEV_TILE_3=14,14 AEV_CFG_3=5+AEV_TILE_3=12,14 | tee $O/ab.txt
for v in "AEV_CFG_3=5" "AEV_CFG_3=5 AEV_TILE_3=12,14"; do env $v AEV_VERBOSE=1 AEV_NO_OVERLAP=1 AEV_STAMPS=3 python bench.py --steps 4 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "AEV_STAMPS\|layer 3:" | grep -v "percentile\|CUs held"; done
