#!/bin/bash
# conv4 / conv5 on the eight-wave instance 9 of the f32 tile kernel against the production instance 4 (interleaved), + its shape tests
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "test_spatial_tile_conv_shapes_vs_oracle and cfg9" 2>&1 | tail -2
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 200"
IFS='|' read -ra SETS <<< "${ENVS:-X=0|AEV_CFG_7=9 AEV_CFG_9=9|X=0|AEV_CFG_7=9 AEV_CFG_9=9}"
for e in "${SETS[@]}"; do env $e python bench.py $Q 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline())
bl=d['roofline']['by_layer']
print('$e', d['ms_per_step'], d['sustained']['ms_per_step'], ' '.join('%s %.1f' % (k, v['us_per_launch']) for k, v in bl.items()))"; done
