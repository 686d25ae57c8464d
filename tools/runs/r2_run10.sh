#!/bin/bash
set -o pipefail
O=gpurun_out/r2_run10; mkdir -p $O
cp async_ev_cnn_amd/libaev_hip.so ab/new.so
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt
tools/ab_bench.sh 2 ab/r1.so ab/new.so ab/new.so+AEV_LOOKAHEAD=0 2>&1 | tee $O/ab.txt
python bench.py > $O/bench.json 2> $O/bench.err; python tools/show_bench.py $O/bench.json | head -12
for S in 64 128; do python bench.py --streams $S --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print($S, d['value'], d['ms_per_step'])"; AEV_LOOKAHEAD=0 python bench.py --streams $S --steps 40 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('noLA', $S, d['value'], d['ms_per_step'])"; done | tee $O/scale.txt
