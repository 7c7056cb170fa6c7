"""Summarise a rocprofv3 *_kernel_stats.csv: top kernels, totals, share of hand-written HIP vs library kernels."""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    n = n.replace("unsigned short", "bf16")
    m = re.match(r"([A-Za-z0-9_:]+(<[^(]*>)?)", n)
    s = m.group(1) if m else n
    return s[:90]


def main(path, div=1.0, top=40):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    ours = 0.0
    print(f"total kernel time {tot/1e6/div:.2f} ms (per unit, div={div})")
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
        t = float(r["TotalDurationNs"])
        print(f"{t/1e6/div:9.3f} ms {100*t/tot:5.1f}%  n={int(r['Calls'])/div:7.1f}  avg={float(r['AverageNs'])/1e3:9.1f} us  {short(r['Name'])}")
    for r in rows:
        if "anonymous namespace" in r["Name"] and "at::native" not in r["Name"]:
            ours += float(r["TotalDurationNs"])
    print(f"hand-written HIP kernels: {100*ours/tot:.1f}% of kernel time")


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0, int(sys.argv[3]) if len(sys.argv) > 3 else 40)
