#!/bin/bash
# Round 3: tile-kernel experiments.  usage: r3_conv.sh <tag> <rounds> variants...   (see tools/ab_bench.sh)
# runs the tile-kernel parity tests with the in-tree library first, then the interleaved A/B (step time + conv class us per
# step), then the in-kernel stamps of conv2 / conv3 for every variant that is a library.
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
T=$1; N=$2; shift 2; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "kernel or efcn or tail or lookahead or config3 or many_streams" > $O/parity.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -4 $O/parity.log
[ $rc -ne 0 ] && exit 1
AB_KERNEL=k_conv_patch tools/ab_bench.sh $N "$@" | tee $O/ab.txt
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in "$@"; do
  if [[ "$v" == *.so ]]; then
    cp $v async_ev_cnn_amd/libaev_hip.so
    for L in 3 5; do AEV_STAMPS=$L python bench.py --steps 6 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense > /dev/null 2> $O/stamps_$(basename $v .so)_$L.txt; echo "$v layer $L"; grep AEV_STAMPS $O/stamps_$(basename $v .so)_$L.txt | tail -2; done
  fi
done
cp $keep async_ev_cnn_amd/libaev_hip.so
