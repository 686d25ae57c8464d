#!/bin/bash
# Round 3: the suite once + the bench line (+ optionally the batch sweep).  usage: r3_check.sh <tag> [sweep]
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/suite.log 2>&1; echo "suite rc=$?"; tail -14 $O/suite.log
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -3 $O/bench.err
python tools/show_bench.py $O/bench.json > $O/bench_summary.txt 2>&1; head -12 $O/bench_summary.txt
if [ "$2" = "sweep" ]; then python bench.py --sweep-batch > $O/sweep.json 2> $O/sweep.err; echo "sweep rc=$?"; cat $O/sweep.json | head -c 3000; fi
if [ "$3" = "stamps" ]; then
  for L in 3 5; do AEV_STAMPS=$L python bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > /dev/null 2> $O/stamps_$L.txt; grep AEV_STAMPS $O/stamps_$L.txt | tail -3; done
fi
