#!/bin/bash
O=gpurun_out/r2_run15; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
AB_KERNEL=k_conv_patch tools/ab_bench.sh 3 AEV_NO_M16=1 - 2>&1 | tee $O/ab.txt
python bench.py --no-cpu-baseline --no-single --no-host-feed > $O/bench.json 2>/dev/null; python tools/show_bench.py $O/bench.json | tail -14
