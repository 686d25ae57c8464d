#!/bin/bash
O=gpurun_out/r2_run23; mkdir -p $O
AEV_FUZZ_SEEDS=150 python -m pytest tests -m gpu -x -q > $O/pytest_fuzz150.log 2>&1; echo "pytest (150 fuzz seeds) rc=$?"; tail -2 $O/pytest_fuzz150.log | cut -c1-200
bash tools/scale_sweep.sh 2>&1 | tee $O/scale.txt
