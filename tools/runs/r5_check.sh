#!/bin/bash
# suite + interleaved A/B against the old k_post + k_post2's phase stamps:  tools/runs/r5_check.sh <tag>
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r05x}
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_suite.log 2>&1; echo "suite rc=$?"; tail -3 gpurun_out/${T}_suite.log
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 0 --profile-steps 0 --steps 40"
AB_SUSTAINED=100 bash tools/ab_bench.sh ${AB_ROUNDS:-2} - AEV_POST_V1=1 2>&1 | tee gpurun_out/${T}_ab.log
AEV_POST_STAMPS=1 python bench.py $Q 2> gpurun_out/${T}_stamps.txt | python tools/show_bench.py /dev/stdin | head -1; grep POST_STAMPS gpurun_out/${T}_stamps.txt | tail -8
AEV_POST_STAMPS=1 python bench.py --streams 1 $Q 2> gpurun_out/${T}_stamps1.txt | python tools/show_bench.py /dev/stdin | head -1; grep POST_STAMPS gpurun_out/${T}_stamps1.txt | tail -8
for s in 1 1; do python bench.py --streams 1 --steps 400 --warmup 20 $Q | python tools/show_bench.py /dev/stdin | head -1; AEV_POST_V1=1 python bench.py --streams 1 --steps 400 --warmup 20 $Q | python tools/show_bench.py /dev/stdin | head -1; done
