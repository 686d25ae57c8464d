#!/bin/bash
O=gpurun_out/r2_run9; mkdir -p $O
for L in 1 3 5; do
  AEV_LOOKAHEAD=1 AEV_LA_BLOCKS=1 AEV_LA_STAMPS=$L python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>&1 >/dev/null | grep LA_STAMPS | tee -a $O/la_stamps.txt
done
AEV_LOOKAHEAD=1 AEV_LA_STAMPS=1 python bench.py --steps 10 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>&1 >/dev/null | grep LA_STAMPS | tee -a $O/la_stamps.txt
