#!/bin/bash
# Round 4, VERDICT item 1: two half batches side by side.  (1) the timing table (1 x 256 against 2 x 128 / 4 x 64 streams, a host thread
# per handle); (2) kernel traces of 1 x 256 and 2 x 128 and what shares the device with what (tools/two_handles_timeline.py).
# -> profiles/r04a_half_batches_*.txt
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=gpurun_out/r4half; mkdir -p $O
[ "$1" = "trace-only" ] || python tools/exp_half_batches.py > $O/half.log 2>&1
cd /tmp && export TMPDIR=/tmp
for G in 1 2; do
  AEV_EXP_TRACE=$G rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace$G -- python3 $R/tools/exp_half_batches.py 256 24 > $R/$O/trace$G.log 2>&1
  python3 $R/tools/two_handles_timeline.py $(find $R/$O/trace$G -name "*kernel_trace.csv" | head -1) "$G handle(s) of $((256 / G)) streams" >> $R/$O/sharing.txt
done
rm -rf $R/$O/trace1 $R/$O/trace2
cat $R/$O/sharing.txt
