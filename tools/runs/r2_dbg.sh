#!/bin/bash
O=gpurun_out/r2_dbg; mkdir -p $O
for i in 1 2 3; do python -m pytest tests -m gpu -x -q > $O/q_$i.log 2>&1; echo "run -q $i rc=$? $(tail -1 $O/q_$i.log | cut -c1-60)"; done
python -m pytest tests -m gpu -x -v > $O/v.log 2>&1; echo "run -v rc=$? $(tail -1 $O/v.log | cut -c1-60)"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-single --no-host-feed 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"
