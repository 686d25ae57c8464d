#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; optionally TCC_HIT_sum + TCC_MISS_sum) into per-kernel HBM
traffic per launch and L2 hit rate.

    python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> out.json [streams] [kind] [l2 csv]

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB, need separate
passes (TCC slots), and on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide (16 B/lane) coalesced streaming
read -- doubled here for k_leak_sweep only, whose loads are exactly that shape (the doubled value then equals its
algorithmic read bytes); narrower / scattered access shapes are uncalibrated and are reported raw."""
import collections
import csv
import json
import sys

WIDE_STREAMING = ("k_leak_sweep",)


def agg(path, ctr):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    f, w, out = agg(sys.argv[1], "FETCH_SIZE"), agg(sys.argv[2], "WRITE_SIZE"), {}
    for k in f:
        if "at::" in k or "rocclr" in k:
            continue
        fetch = f[k][1] / f[k][0] * 1024.0
        write = (w[k][1] / w[k][0] * 1024.0) if k in w else 0.0
        corr = 2.0 if k.startswith(WIDE_STREAMING) else 1.0
        out[k] = {"launches": f[k][0], "fetch_bytes_raw": round(fetch), "fetch_correction": corr,
                  "write_bytes": round(write), "hbm_bytes_per_launch": round(fetch * corr + write)}
    if len(sys.argv) > 6:  # L2 hit rate: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)  (MI355X_MICROARCH.md, L2)
        hit, miss = agg(sys.argv[6], "TCC_HIT_sum"), agg(sys.argv[6], "TCC_MISS_sum")
        for k in out:
            if k in hit and k in miss and hit[k][1] + miss[k][1] > 0:
                out[k]["l2_hit_rate"] = round(hit[k][1] / (hit[k][1] + miss[k][1]), 4)
    meta = {"streams_per_gpu": int(sys.argv[4]) if len(sys.argv) > 4 else None,
            "stream_kind": sys.argv[5] if len(sys.argv) > 5 else "ncaltech", "kernels": out,
            "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; KiB -> bytes; FETCH_SIZE doubled for the "
                    "16-B/lane streaming kernel only (MI355X_MICROARCH.md, HBM)"}
    json.dump(meta, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"]):
        print("%-44s %5d launches  %8.1f MB/launch  L2 hit %s" % (k[:44], v["launches"], v["hbm_bytes_per_launch"] / 1e6, v.get("l2_hit_rate")))


if __name__ == "__main__":
    main()
