#!/bin/bash
O=gpurun_out/r2_run14; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
tools/ab_bench.sh 2 ab/head.so ab/new.so 2>&1 | tee $O/ab.txt
