#!/bin/bash
# timeline of replayed steps on another stream kind: r3_tl_kind.sh <tag> <kind> <batch> <streams> [env...]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
T=$1; K=$2; B=$3; S=$4; shift 4
O=gpurun_out/$T; mkdir -p $R/$O
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/stats -- python3 $R/bench.py --kind $K --batch $B --streams $S --steps 24 --warmup 8 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > $R/$O/stats.log 2>&1
cd $R
python tools/timeline.py $(find $O/stats -name "*kernel_trace.csv" | head -1) 7 1 > $O/timeline.txt
rm -rf $O/stats
cat $O/timeline.txt
