"""Rounds of resident workgroups per kernel of one train step: from a rocprofv3 --kernel-trace CSV (grid, workgroup size, VGPRs, LDS per dispatch)
estimate the workgroups a CU holds (wave slots by VGPRs: 512 / vgpr per SIMD, <= 8; LDS: 160 KB; 2048 threads) and print workgroups / (slots x CUs).
A fraction just above a whole number means a last round that runs nearly empty.   python tools/rounds.py <kernel_trace.csv> [steps]"""
import csv
import sys
from collections import defaultdict

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
CUS = 256
agg = defaultdict(lambda: {"n": 0, "us": 0.0, "wgs": 0, "wg_size": 0, "vgpr": 0, "lds": 0})
with open(path) as fh:
    for r in csv.DictReader(fh):
        name = r.get("Kernel_Name", "")
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", "0")) or 0) * max(1, int(r.get("Grid_Size_Y", "1") or 1)) * max(1, int(r.get("Grid_Size_Z", "1") or 1))
        wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1")) or 1) * max(1, int(r.get("Workgroup_Size_Y", "1") or 1)) * max(1, int(r.get("Workgroup_Size_Z", "1") or 1))
        vg = 2 * (int(r.get("VGPR_Count", r.get("Arch_VGPR_Count", "0")) or 0) + int(r.get("Accum_VGPR_Count", "0") or 0))   # the trace counts per 32 lanes (214 registers -> 108)
        lds = int(r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "0")) or 0)
        t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
        key = (name, gx // max(wx, 1), wx, vg, lds)
        a = agg[key]
        a["n"] += 1
        a["us"] += t
rows = []
for (name, wgs, wx, vg, lds), a in agg.items():
    waves = (wx + 63) // 64
    per_simd = min(8, 512 // max(vg, 1)) if vg else 8
    slots_v = (per_simd * 4) // waves if waves else 0
    slots_l = (160 * 1024) // lds if lds else 99
    slots_t = 2048 // wx if wx else 99
    slots = max(1, min(slots_v, slots_l, slots_t, 32))
    rounds = wgs / (slots * CUS)
    rows.append((a["us"] / steps, name, a["n"] / steps, a["us"] / a["n"], wgs, wx, vg, lds, slots, rounds))
rows.sort(reverse=True)
print(f"{'ms/step':>8s} {'n/step':>6s} {'avg us':>8s} {'wgs':>6s} {'thr':>4s} {'vgpr':>4s} {'lds':>6s} {'wg/CU':>5s} {'rounds':>7s}  kernel")
for ms, name, n, avg, wgs, wx, vg, lds, slots, rounds in rows[:60]:
    nm = name.replace("unsigned short", "bf16").replace("(anonymous namespace)::", "").split("(")[0]
    print(f"{ms * 1e-3:8.3f} {n:6.1f} {avg:8.1f} {wgs:6d} {wx:4d} {vg:4d} {lds:6d} {slots:5d} {rounds:7.2f}  {nm[:90]}")
