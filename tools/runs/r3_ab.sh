#!/bin/bash
# usage: r3_ab.sh <tag> <rounds> <variant> <variant> ...   (tools/ab_bench.sh variants; AB_KERNEL honoured)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; N=$2; shift; shift
O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/suite.log 2>&1; echo "suite rc=$?"; tail -3 $O/suite.log
bash tools/ab_bench.sh $N "$@" 2>&1 | tee $O/ab.log
