#!/bin/bash
# kernel trace of a short queued run: where the fixed cost of a run goes (docs/KERNEL_NOTES.md, round 4)
R=${GRAFT_REPO_ROOT:-$(git rev-parse --show-toplevel)}
cd "$R" || exit 1
O=$R/gpurun_out/fc; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
AEV_FC_PLAN="20,3,3,3" rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $R/tools/exp_run_fixed_cost.py > $O/log.txt 2>&1
cd "$R" || exit 1; python - <<'PY'
import csv, glob, re
f = glob.glob("gpurun_out/fc/tr/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
peeks = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_integ_peek")]
a, b = peeks[-2], peeks[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = int(rows[a - 1]["End_Timestamp"])
print("gap before the run's first kernel: %.1f us" % ((t0 - prev_end) / 1e3))
out = open("gpurun_out/fc/run_of_3_steps.txt", "w")
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    out.write("%9.1f %8.1f  q%s  %-40s %sx%s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r["Queue_Id"], nm[:40], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]))
last = max(int(r["End_Timestamp"]) for r in rows[a:b])
print("run of 3 steps: first kernel to last kernel end %.1f us, %d kernels" % ((last - t0) / 1e3, b - a))
PY
rm -rf $O/tr
