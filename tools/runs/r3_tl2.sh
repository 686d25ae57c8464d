#!/bin/bash
# timelines of two libraries: r3_tl2.sh libA libB
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
keep=$(mktemp); cp async_ev_cnn_amd/libaev_hip.so $keep
for v in "$@"; do
  cp $v async_ev_cnn_amd/libaev_hip.so
  t=tl_$(basename $v .so)
  tools/runs/r3_timeline.sh $t 256 > /dev/null 2>&1
  echo "== $v"; head -1 gpurun_out/$t/timeline_S256.txt; grep -E "k_conv_patch|k_conv_1x1|k_post|k_pool|k_leak|k_conv_first|k_integ" gpurun_out/$t/timeline_S256.txt | head -45
done
cp $keep async_ev_cnn_amd/libaev_hip.so
