#!/bin/bash
# events/s and ms per step by streams per GPU
for S in 1 4 16 32 64 128 160 192 256 512 1024; do python bench.py --streams $S --steps 100 --warmup 10 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print($S, d['value'], d['ms_per_step'])"; done
