#!/bin/bash
for L in 3 5; do
AEV_STAMPS=$L python bench.py --steps 6 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>&1 >/dev/null | grep AEV_STAMPS | tail -3
done
