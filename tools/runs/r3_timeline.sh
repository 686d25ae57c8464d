#!/bin/bash
# timeline of replayed steps (rocprofv3 kernel trace): tools/runs/r3_timeline.sh <tag> <streams> [env...]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
T=${1:-tl}; S=${2:-256}; shift; shift
O=gpurun_out/$T; mkdir -p $R/$O
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --streams $S --steps 24 --warmup 8 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > $R/$O/stats.log 2>&1
cd $R
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S$S.txt
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_S$S.csv
rm -rf $O/stats
cat $O/timeline_S$S.txt
