#!/bin/bash
# confidence run: the GPU fuzzers with 120 seeds each, several times (the initialisation race of round 1 showed up once in ~1500 nets)
O=gpurun_out/r2_fuzz; mkdir -p $O
grep -h unique_id /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | sort -u | tail -1
for i in 1 2 3 4 5 6 7 8; do AEV_FUZZ_SEEDS=120 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py -m gpu -q -k "random_arch" > $O/fz_$i.log 2>&1; echo "fuzz run $i rc=$? $(tail -1 $O/fz_$i.log | cut -c1-70)"; grep -h "AssertionError:" $O/fz_$i.log | cut -c1-700; done
