#!/bin/bash
O=gpurun_out/r2_run24; mkdir -p $O
AEV_EARLY_FORK=1 python -m pytest tests -m gpu -x -q -k "efcn_128 or lookahead or many_streams" 2>&1 | tail -1
tools/ab_bench.sh 4 - AEV_EARLY_FORK=1 2>&1 | tee $O/ab.txt
