#!/bin/bash
O=gpurun_out/r2_run13; mkdir -p $O
tools/ab_bench.sh 4 ab/head.so ab/new.so 2>&1 | tee $O/ab.txt
