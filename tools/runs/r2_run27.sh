#!/bin/bash
O=gpurun_out/r2_run27; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-100
AB_KERNEL=k_pool_eval tools/ab_bench.sh 3 AEV_POOL_U=1 - 2>&1 | tee $O/ab.txt
