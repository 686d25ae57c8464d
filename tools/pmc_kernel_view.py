#!/usr/bin/env python3
"""Per-kernel means of one or more rocprofv3 --pmc counter_collection.csv files (diagnostics).

    python tools/pmc_kernel_view.py <substring of kernel name> file1.csv [file2.csv ...]"""
import collections
import csv
import sys


def main():
    pat, files = sys.argv[1], sys.argv[2:]
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            k = k.split("(")[0].replace("void ", "")
            d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in sorted(d.items()):
        print(k)
        for c, x in sorted(v.items()):
            print("    %-36s %14.0f  (%d launches)" % (c, sum(x) / len(x), len(x)))


if __name__ == "__main__":
    main()
