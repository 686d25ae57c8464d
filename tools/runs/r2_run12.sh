#!/bin/bash
O=gpurun_out/r2_run12; mkdir -p $O
cp async_ev_cnn_amd/libaev_hip.so ab/new2.so
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $O/pytest.log | cut -c1-100
tools/ab_bench.sh 3 ab/new.so ab/new2.so 2>&1 | tee $O/ab.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/timeline.py $(find $O/prof -name "*kernel_trace.csv" | head -1) 3 | tee $O/timeline.txt
