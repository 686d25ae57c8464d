"""How do the kernels of G handles share the device?  From a rocprofv3 --kernel-trace CSV of tools/exp_half_batches.py (AEV_EXP_TRACE=<G>):
over the steady part of the run, the share of the wall time with a tile kernel (k_conv_patch, MFMA-bound) resident, with TWO OR MORE
tile kernels resident at once, with only other kernels resident, and with nothing resident; and how long the non-tile kernels take
when they start while a tile kernel is resident against when they have the device to themselves.
    python tools/two_handles_timeline.py <kernel_trace.csv> [label]"""
import csv
import re
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if r["Kind"] == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
label = sys.argv[2] if len(sys.argv) > 2 else ""
name = lambda r: re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]
# steady part: the last 60 % of the k_integ_prepare launches
prep = [i for i, r in enumerate(rows) if name(r) == "k_integ_prepare"]
a = prep[int(len(prep) * 0.4)]
b = prep[-1]
win = rows[a:b]
t0, t1 = int(win[0]["Start_Timestamp"]), int(rows[b]["Start_Timestamp"])
ev = []
for r in win:
    s, e = int(r["Start_Timestamp"]), min(int(r["End_Timestamp"]), t1)
    tile = name(r) == "k_conv_patch"
    ev.append((s, 1, tile)); ev.append((e, -1, tile))
ev.sort()
nt = nl = 0
last = t0
acc = {"tile>=2": 0, "tile==1": 0, "light only": 0, "idle": 0, "tile + light": 0}
for t, d, tile in ev:
    dt = t - last
    if dt > 0:
        if nt >= 2: acc["tile>=2"] += dt
        elif nt == 1: acc["tile==1"] += dt
        elif nl > 0: acc["light only"] += dt
        else: acc["idle"] += dt
        if nt >= 1 and nl > 0: acc["tile + light"] += dt
    last = t
    if tile: nt += d
    else: nl += d
tot = float(t1 - t0)
steps = len([i for i in prep if a <= i < b])
print("%s: %d k_integ_prepare launches in %.2f ms (%.1f us per launch)" % (label, steps, tot / 1e6, tot / 1e3 / max(steps, 1)))
for k in ("tile>=2", "tile==1", "light only", "idle"):
    print("  %-12s %5.1f %% of the wall time" % (k, 100 * acc[k] / tot))
print("  (a non-tile kernel resident beside a tile kernel: %.1f %%)" % (100 * acc["tile + light"] / tot))
# duration of the non-tile kernels by whether a tile kernel was resident when they started
tiles = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in win if name(r) == "k_conv_patch")
import bisect
starts = [s for s, _ in tiles]
by = {}
for r in win:
    nm = name(r)
    if nm == "k_conv_patch":
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    i = bisect.bisect_right(starts, s) - 1
    beside = any(ts <= s < te for ts, te in tiles[max(0, i - 8):i + 1])
    by.setdefault((nm, beside), []).append((e - s) / 1e3)
print("  mean duration (us) of the other kernels: started while a tile kernel was resident / otherwise")
for nm in sorted(set(k[0] for k in by)):
    x, y = by.get((nm, True), []), by.get((nm, False), [])
    print("    %-22s %8s (%4d launches) / %8s (%4d)" % (nm, "%.1f" % (sum(x) / len(x)) if x else "-", len(x), "%.1f" % (sum(y) / len(y)) if y else "-", len(y)))
