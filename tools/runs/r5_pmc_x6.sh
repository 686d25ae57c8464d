#!/bin/bash
# Instruction counts, MFMA busy cycles, waits and LDS conflicts of the split-bf16 tile kernel (eager launches), separate PMC passes.
#   usage: tools/runs/r5_pmc_x6.sh <tag>   -> gpurun_out/<tag>/p*.txt
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=gpurun_out/${1:-r5pmcx6}; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --conv-math bf16x6 --steps 4 --warmup 2 --profile-steps 0 --sustained-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-x6 --no-graph"
i=0
for G in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$O/p$i -- $B > $R/$O/p$i.log 2>&1
  f=$(find $R/$O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_kernel_view.py k_conv_x6 $f > $R/$O/p$i.txt 2>&1; else tail -3 $R/$O/p$i.log; fi
  rm -rf $R/$O/p$i
done
cd $R; cat $O/p*.txt
