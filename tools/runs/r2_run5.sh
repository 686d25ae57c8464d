#!/bin/bash
# round 2, GPU run 5: baseline timeline of a replayed step (kernel trace)
set -o pipefail
O=gpurun_out/r2_run5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
AEV_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_noov -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof_noov.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt
python tools/timeline.py $(find $O/prof_noov -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline_noov.txt
