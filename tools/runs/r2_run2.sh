#!/bin/bash
# round 2, GPU run 2: a persistent streaming kernel beside the deep conv layers (dummy buffers): what does co-running cost?
set -o pipefail
O=gpurun_out/r2_run2; mkdir -p $O
tools/ab_bench.sh 2 - AEV_EXP_COSWEEP=1,400,7 AEV_EXP_COSWEEP=2,400,7 AEV_EXP_COSWEEP=4,400,7 AEV_EXP_COSWEEP=2,400,7+AEV_EXP_PRIO=1 AEV_EXP_COSWEEP=2,400,12 AEV_EXP_COSWEEP=8,400,12 AEV_EXP_COSWEEP=2,400,5 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
AEV_EXP_COSWEEP=2,400,7 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/show_stats.py $(find $O/prof -name "*kernel_stats.csv" | head -1) | head -30 | tee $O/stats.txt
