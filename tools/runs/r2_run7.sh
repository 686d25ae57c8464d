#!/bin/bash
# round 2, GPU run 6: sweeps with unconditional loads; pipelined look-ahead kernel
set -o pipefail
O=gpurun_out/r2_run7; mkdir -p $O
cp async_ev_cnn_amd/libaev_hip.so ab/new.so
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest.log | tee -a $O/summary.txt

tools/ab_bench.sh 2 ab/new.so ab/new.so+AEV_LOOKAHEAD=1 ab/new.so+AEV_LOOKAHEAD=1+AEV_LA_PRIO=1 ab/new.so+AEV_LOOKAHEAD=1+AEV_LA_BLOCKS=1 ab/new.so+AEV_LOOKAHEAD=1+AEV_LA_BLOCKS=2 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
AEV_LOOKAHEAD=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt
