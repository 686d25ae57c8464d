#!/bin/bash
O=gpurun_out/r2_run19; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-300
AEV_FUZZ_SEEDS=60 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py -m gpu -x -q -k "random_arch" > $O/fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
