#!/bin/bash
# the x6 tests + bench runs in split-bf16 mode (headline ms, sustained ms, eager us per launch of the conv layers):  [ENVS="A=1 B=2|C=3"] tools/runs/r5_x6_quick.sh
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_x6.py -x -q 2>&1 | tail -2
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 200"
IFS='|' read -ra SETS <<< "${ENVS:-X=0|X=0}"
for e in "${SETS[@]}"; do env $e python bench.py $Q --conv-math bf16x6 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
bl=d['roofline']['by_layer']
print('$e', d['ms_per_step'], d['sustained']['ms_per_step'], ' '.join('%s %.1f' % (k, v['us_per_launch']) for k, v in bl.items()))"; done
