#!/bin/bash
# One replayed 256-stream step (and one 1-stream step) as a timeline + kernel stats under rocprofv3:  tools/runs/r5_timeline.sh <tag> [env ...]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r05t}; shift
O=$R/gpurun_out/$T; mkdir -p $O
for e in "$@"; do export "$e"; done
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 0 --profile-steps 0"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 24 --warmup 8 $Q > $O/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -- python3 $R/bench.py --streams 1 --steps 24 --warmup 8 $Q > $O/stats1.log 2>&1
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_S256.csv
cp $(find $O/stats1 -name "*kernel_stats.csv" | head -1) $O/kernel_stats_S1.csv
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S256.txt
python tools/timeline.py $(find $O/stats1 -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S1.txt
rm -rf $O/stats $O/stats1
head -50 $O/timeline_S256.txt
