"""Per-kernel register / LDS / occupancy table from hipcc's -Rpass-analysis=kernel-resource-usage (cross-compiles, no GPU needed).
Usage: python tools/kres.py [file.hip ...]   (default: every csrc/*.hip)"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "pixelspointspolygons_amd/csrc/*.hip")))
for f in files:
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-c", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
                        f, "-o", "/dev/null"], capture_output=True, text=True)
    cur = None
    rows = []
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            name = subprocess.run(["c++filt", t.split(": ")[1]], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(.*$", "", name.replace("(anonymous namespace)::", "").replace("void ", "")).replace("unsigned short", "bf16")
            cur = {"name": name}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    print(f"== {os.path.basename(f)}")
    for c in rows:
        print(f"  v={c.get('VGPRs', '?'):>4} a={c.get('AGPRs', '?'):>4} occ={c.get('Occupancy [waves/SIMD]', '?'):>2} scratch={c.get('ScratchSize [bytes/lane]', '?'):>5} "
              f"lds={c.get('LDS Size [bytes/block]', '?'):>6}  {c['name'][:100]}")
