#!/bin/bash
O=gpurun_out/r2_run17; mkdir -p $O
python -m pytest tests -m gpu -x -q -k "lookahead or efcn_128 or baseline_size" 2>&1 | tail -1
tools/ab_bench.sh 2 - AEV_LA_BLOCKS=2 AEV_LA_BLOCKS=2+AEV_NO_M16=1 AEV_NO_M16=1 AEV_LA_BLOCKS=3 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
AEV_LA_BLOCKS=2 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt
