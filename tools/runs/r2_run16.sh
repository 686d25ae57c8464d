#!/bin/bash
O=gpurun_out/r2_run16; mkdir -p $O
tools/ab_bench.sh 2 AEV_NO_M16=1 - AEV_LA_PRIO=1 AEV_LA_MAX=2 AEV_LA_MAX=1 AEV_LOOKAHEAD=0 AEV_LOOKAHEAD=0+AEV_NO_M16=1 AEV_LA_MAX=2+AEV_LA_PRIO=1 2>&1 | tee $O/ab.txt
