"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE) per kernel: mean HBM bytes per launch.

Usage: python tools/pmc_traffic.py <dir with *_counter_collection.csv of the FETCH pass> <dir of the WRITE pass> <out.json> [train steps profiled]
With the step count the file also carries "_steps" and "_step_total_bytes" (what bench.py reports as roofline.step_traffic_bytes).
Units / corrections follow /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE are reported in KiB;
on gfx950 FETCH_SIZE counts 128-byte requests at 64 B, so wide coalesced reads are doubled (`fetch_bytes_corrected`);
WRITE_SIZE is uncalibrated and reported as is.
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def collect(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = re.sub(r"\(.*$", "", row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
                a = acc[name]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    fd, wd, out = sys.argv[1:4]
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    fe, wr = collect(fd, "FETCH_SIZE"), collect(wd, "WRITE_SIZE")
    res = {}
    for k in sorted(set(fe) | set(wr), key=lambda k: -(fe.get(k, [0, 0])[0] + wr.get(k, [0, 0])[0])):
        f, w = fe.get(k, [0.0, 0]), wr.get(k, [0.0, 0])
        res[k] = {"launches": max(f[1], w[1]),
                  "fetch_bytes_raw": f[0] * 1024 / max(f[1], 1), "fetch_bytes_corrected": 2 * f[0] * 1024 / max(f[1], 1),
                  "write_bytes": w[0] * 1024 / max(w[1], 1)}
    if steps:
        res["_steps"] = steps
        res["_step_total_bytes"] = sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in res.values() if isinstance(v, dict)) / steps
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for k, v in [kv for kv in res.items() if isinstance(kv[1], dict)][:25]:
        print(f"{v['launches']:6d}  fetch {v['fetch_bytes_corrected'] / 1e6:9.2f} MB  write {v['write_bytes'] / 1e6:9.2f} MB  {k[:90]}")


if __name__ == "__main__":
    main()
