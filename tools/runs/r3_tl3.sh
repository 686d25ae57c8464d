#!/bin/bash
# kernel stats (graph replays) of libraries at a stream count: r3_tl3.sh <streams> libs...
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
S=$1; shift
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in "$@"; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  t=tls_$(basename $v .so)
  tools/runs/r3_timeline.sh $t $S > /dev/null 2>&1
  echo "== $v"; head -1 gpurun_out/$t/timeline_S$S.txt
  python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/$t/kernel_stats_S$S.csv')))
for r in rows[:14]:
    print("%-60s calls %6s avg %9.1f us total %10.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e3))
PY
done
cp $keep async_ev_cnn_amd/libaev_hip.so
