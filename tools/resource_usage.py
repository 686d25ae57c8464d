"""Per-kernel register / LDS / scratch usage of the library (dev helper):
    hipcc ... -Rpass-analysis=kernel-resource-usage aev_hip.hip 2> log ; python tools/resource_usage.py log [filter]"""
import re
import subprocess
import sys


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            out[cur] = {}
            continue
        m = re.search(r"remark:\s+([^:]+?): (\S+)", line)
        if m and cur:
            out[cur][m.group(1).strip()] = m.group(2)
    return out


def main():
    d = parse(sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    names = list(d)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    for k, nm in zip(names, dem):
        if flt in nm:
            v = d[k]
            nm = re.sub(r"^void ", "", nm).split("(")[0]
            print("%-44s VGPR %-4s AGPR %-4s spill %-3s scratch %-5s occ %-2s LDS %s" % (
                nm[:44], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill", v.get("VGPR Spill")), v.get("ScratchSize [bytes/lane]", v.get("ScratchSize")),
                v.get("Occupancy [waves/SIMD]", v.get("Occupancy")), v.get("LDS Size [bytes/block]", v.get("LDS Size"))))


if __name__ == "__main__":
    main()
