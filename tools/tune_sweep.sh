#!/bin/bash
# ms per step by stream count and kernel configuration policy: tools/tune_sweep.sh "16 32 64 128"
for S in ${1:-16 32 64 128}; do for T in auto throughput latency; do
  ms=$(python bench.py --streams $S --tuning $T --steps 100 --warmup 10 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.readline())['ms_per_step'])")
  echo "S=$S tuning=$T $ms"
done; done
