#!/bin/bash
# round 2, GPU run 1: test-suite, the stale-flag test against the round-1 library, rotation A/B, stamps
set -o pipefail
O=gpurun_out/r2_run1; mkdir -p $O
cp async_ev_cnn_amd/libaev_hip.so ab/new.so
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
cp ab/r1.so async_ev_cnn_amd/libaev_hip.so
python -m pytest tests/test_gpu_parity.py -m gpu -q -k mixed_layer_api > $O/pytest_mixed_r1lib.log 2>&1; echo "mixed test on the round-1 library rc=$? (expected to fail without flags env)" | tee -a $O/summary.txt
cp ab/new.so async_ev_cnn_amd/libaev_hip.so
tools/ab_bench.sh 3 ab/r1.so ab/new.so ab/new.so+AEV_NO_ROTATE=1 2>&1 | tee $O/ab_rotate.txt
python bench.py --no-cpu-baseline --no-single --no-host-feed > $O/bench_new.json 2> $O/bench_new.err
AEV_NO_ROTATE=1 python bench.py --no-cpu-baseline --no-single --no-host-feed > $O/bench_norot.json 2> $O/bench_norot.err
for L in 3 5 7 9; do
  AEV_STAMPS=$L python bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > /dev/null 2> $O/stamps_L$L.txt
  AEV_NO_ROTATE=1 AEV_STAMPS=$L python bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > /dev/null 2> $O/stamps_L${L}_norot.txt
done
grep -h AEV_STAMPS $O/stamps_L*.txt | tail -40
