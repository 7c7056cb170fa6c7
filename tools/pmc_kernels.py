"""Per-kernel mean of every counter found in rocprofv3 *_counter_collection.csv files under a directory.
Usage: python tools/pmc_kernels.py <dir> [name-filter-regex]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")).replace("unsigned short", "bf16")
            if flt and not flt.search(name):
                continue
            a = acc[name][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
for name, cs in sorted(acc.items(), key=lambda kv: -max(v[0] for v in kv[1].values())):
    n = max(v[1] for v in cs.values())
    print(f"{name[:80]}  (n={n})")
    wc = cs.get("SQ_WAVE_CYCLES", [0, 1])
    wcm = wc[0] / max(wc[1], 1)
    for c, (s, k) in sorted(cs.items()):
        m = s / max(k, 1)
        extra = f"  {m / wcm:6.1%} of SQ_WAVE_CYCLES" if wcm and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else ""
        print(f"    {c:32s} {m:16.1f}{extra}")
