#!/bin/bash
# timeline of one replayed step (rocprofv3 kernel trace of the graph replay): tools/runs/r2_timeline.sh <tag> [env...]
T=${1:-tl}; shift
O=gpurun_out/$T; mkdir -p $O
R=$GRAFT_REPO_ROOT
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --steps 24 --warmup 8 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $R/$O/stats.log 2>&1
cd $R
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 4 > $O/timeline_S256.txt
rm -rf $O/stats
