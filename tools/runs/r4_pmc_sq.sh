#!/bin/bash
# Instruction counts and MFMA busy cycles of the tile-kernel instances (eager per-kernel pass), two PMC passes.
#   usage: tools/runs/r4_pmc_sq.sh <tag>   -> gpurun_out/<tag>/p1.txt, p2.txt  (profiles/<tag>_pmc_sq_instruction_counts_tile_kernels.txt)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=gpurun_out/${1:-r4pmc}; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --sustained-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-graph"
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$O/p$i -- $B > $R/$O/p$i.log 2>&1
  f=$(find $R/$O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_kernel_view.py k_conv_patch $f > $R/$O/p$i.txt 2>&1; fi
  rm -rf $R/$O/p$i
done
cd $R; cat $O/p1.txt $O/p2.txt
