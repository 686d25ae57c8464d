#!/bin/bash
# exploratory PMC passes over the eager per-kernel pass (conv kernels), one counter group per run
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=gpurun_out/r3pmc; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/$O/avail.txt 2>&1
B="python3 $R/bench.py --steps 4 --warmup 2 --profile-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-graph"
i=0
for G in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$O/p$i -- $B > $R/$O/p$i.log 2>&1
  f=$(find $R/$O/p$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 $R/tools/pmc_kernel_view.py k_conv_patch $f > $R/$O/p$i.txt 2>&1; fi
  rm -rf $R/$O/p$i
done
cd $R; for f in $O/p*.txt; do echo "== $f"; head -30 $f; done
