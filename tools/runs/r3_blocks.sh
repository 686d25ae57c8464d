#!/bin/bash
# Round 3: how many blocks of conv2's tile kernel are resident per CU?  AEV_BLOCKS_3 = persistent blocks per CU.
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/blocks; mkdir -p $O
AB_KERNEL=k_conv_patch tools/ab_bench.sh 2 - AEV_BLOCKS_3=2 AEV_BLOCKS_3=4 AEV_BLOCKS_5=1 AEV_BLOCKS_5=3 | tee $O/ab.txt
for B in 2 3 4; do AEV_BLOCKS_3=$B AEV_STAMPS=3 python bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > /dev/null 2> $O/stamps_b$B.txt; echo "blocks $B"; grep AEV_STAMPS $O/stamps_b$B.txt | tail -3; done
AEV_VERBOSE=1 python bench.py --steps 2 --warmup 1 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense 2>&1 >/dev/null | grep "tile kernel" | head
