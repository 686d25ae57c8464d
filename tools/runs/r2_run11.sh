#!/bin/bash
O=gpurun_out/r2_run11; mkdir -p $O
AEV_VERBOSE=1 python bench.py --steps 3 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>&1 >/dev/null | grep "tile kernel" | tee $O/cfg.txt
AB_KERNEL=k_conv_patch tools/ab_bench.sh 2 - AEV_CFG_3=5+AEV_TILE_3=14,14 AEV_CFG_3=5+AEV_TILE_3=10,14 AEV_CFG_3=5+AEV_TILE_3=12,14 AEV_CFG_3=5+AEV_TILE_3=8,14 2>&1 | tee $O/ab.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "efcn_whole_chain" 2>&1 | tail -1
AEV_CFG_3=5 AEV_TILE_3=14,14 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "efcn_whole_chain or efcn_128" 2>&1 | tail -1
