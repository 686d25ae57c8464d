"""One step of the replayed graph as a timeline, from a rocprofv3 --kernel-trace CSV (dev helper):
    python tools/timeline.py <kernel_trace.csv> [step index from the end, default 3] [steps to print, default 1]"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("k_integ_prepare", "k_integ_step", "k_head"))]  # the first launch of a step on the chain (k_head: up to four streams)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if len(starts) < 2:
    sys.exit("timeline.py: no step start (k_integ_prepare / k_integ_step / k_head) in %s" % sys.argv[1])
m = int(sys.argv[3]) if len(sys.argv) > 3 else 1
a, b = starts[-k - 1], starts[-k - 1 + m]
t0 = int(rows[a]["Start_Timestamp"])
last_end = {}
busy_end = t0
print("step of %.1f us, %d kernels" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, b - a))
print("%9s %9s %8s  q  %-46s %s" % ("start us", "dur us", "gap us", "kernel", "grid"))
for r in rows[a:b]:
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]
    gap = (s - busy_end) / 1e3          # time with NOTHING running on the device before this kernel starts
    nm = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
    print("%9.1f %9.1f %8.1f  %s  %-46s %sx%s" % ((s - t0) / 1e3, (e - s) / 1e3, max(gap, 0.0), q, nm[:46], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), r["Grid_Size_Y"]))
    busy_end = max(busy_end, e)
