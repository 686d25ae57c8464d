#!/bin/bash
O=gpurun_out/r2_run25; mkdir -p $O
python bench.py --steps 1500 --warmup 5 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $O/long.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor (edge|junction)" | tr '\n' ';'; echo; sleep 0.5; done | tee $O/smi.txt
wait $BP
python -c "import json; d=json.loads(open('$O/long.json').readline()); print(d['value'], d['ms_per_step'])"
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ';'; echo
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
