#!/bin/bash
# round 2, GPU run 3: look-ahead leak sweeps: parity + A/B
set -o pipefail
O=gpurun_out/r2_run3; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest.log | tee -a $O/summary.txt
tools/ab_bench.sh 2 AEV_NO_LOOKAHEAD=1 - AEV_LA_BLOCKS=2 AEV_LA_BLOCKS=4 AEV_PRIO=1 AEV_LA_BLOCKS=2+AEV_PRIO=1 2>&1 | tee $O/ab.txt
python bench.py --no-cpu-baseline --no-single --no-host-feed > $O/bench_new.json 2> $O/bench_new.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
cat $(find $O/prof -name "*kernel_stats.csv" | head -1) | cut -c1-150 | head -30 | tee $O/stats.txt
