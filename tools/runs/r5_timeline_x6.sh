#!/bin/bash
# One replayed 256-stream step in split-bf16 mode as a timeline + kernel stats under rocprofv3:  tools/runs/r5_timeline_x6.sh <tag> [env ...]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r05x}; shift
O=$R/gpurun_out/$T; mkdir -p $O
for e in "$@"; do export "$e"; done
Q="--no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --sustained-steps 0 --profile-steps 0 --conv-math bf16x6"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 24 --warmup 8 $Q > $O/stats.log 2>&1
cd $R
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/kernel_stats_S256_bf16x6.csv
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 2 > $O/timeline_S256_bf16x6.txt
rm -rf $O/stats
head -60 $O/timeline_S256_bf16x6.txt
