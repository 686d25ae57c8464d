#!/bin/bash
# the x6 tests + two bench runs in split-bf16 mode (headline ms, sustained ms, eager us per launch of the conv layers)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
timeout -k 10 300 python -m pytest tests/test_gpu_x6.py -x -q 2>&1 | tail -2
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 200"
for m in ${MODES:-bf16x6 bf16x6}; do python bench.py $Q --conv-math $m 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
bl=d['roofline']['by_layer']
print('$m', d['ms_per_step'], d['sustained']['ms_per_step'], ' '.join('%s %.1f' % (k, v['us_per_launch']) for k, v in bl.items()))"; done
