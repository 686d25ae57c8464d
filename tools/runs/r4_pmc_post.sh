#!/bin/bash
# k_post launch by launch (one stream by default): instruction counts, wave cycles, waits -- where do a launch's microseconds go?
#   usage: tools/runs/r4_pmc_post.sh <tag> [streams]   -> gpurun_out/<tag>/post_pmc.txt
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
O=gpurun_out/${1:-r4post}; S=${2:-1}; mkdir -p $R/$O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --streams $S --steps 4 --warmup 2 --profile-steps 0 --sustained-steps 0 --no-cpu-baseline --no-single --no-host-feed --no-dense --no-legacy --no-graph"
i=0
: > $R/$O/post_pmc.txt
for G in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $G --kernel-trace --output-format csv -d $R/$O/q$i -- $B > $R/$O/q$i.log 2>&1
  f=$(find $R/$O/q$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python3 - "$f" >> $R/$O/post_pmc.txt <<'PY'
import csv, sys, collections
rows = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    if not r["Kernel_Name"].startswith("k_post"): continue
    rows.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
ids = sorted(rows)
names = sorted({c for v in rows.values() for c in v})
print("dispatch " + " ".join("%22s" % c for c in names))
for d in ids[-16:]:      # the last two steps' launches (8 per e-fcn step)
    print("%8d " % d + " ".join("%22.0f" % rows[d].get(c, 0) for c in names))
PY
  fi
  rm -rf $R/$O/q$i
done
cd $R; cat $O/post_pmc.txt
