#!/bin/bash
O=gpurun_out/r2_run18; mkdir -p $O
tools/ab_bench.sh 3 ab/head.so ab/it3.so ab/it2.so ab/it2.so+AEV_LA_BLOCKS=2 ab/it3.so+AEV_NO_M16=1 2>&1 | tee $O/ab.txt
