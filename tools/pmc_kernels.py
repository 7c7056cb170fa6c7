"""Per-kernel mean of every counter found in rocprofv3 *_counter_collection.csv files under a directory.
Usage: python tools/pmc_kernels.py <dir> [name-filter-regex]

Units (MI355X_MICROARCH.md, "s_memtime tick vs SQ PMC units"): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_LDS_* count QUAD-cycles summed over waves,
SQ_VALU_MFMA_BUSY_CYCLES counts CYCLES (32 per 32x32x16 bf16 MFMA) summed over SIMDs.  The percentage column is against SQ_WAVE_CYCLES in the same unit
(MFMA busy / 4), i.e. the share of a resident wave's time; "per-SIMD pipe" = MFMA busy cycles / (kernel cycles x SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs,
1024 SIMDs: the matrix-pipe utilisation of the chip while the kernel runs (r05's summaries printed cycles over quad-cycles: 4x too high).
r06: SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT count LDS-array CYCLES summed over CUs (checked on the dK/dV attention kernel: its instruction count x the guide's
cycles per LDS instruction gives the counter to 20 %), not quad-cycles of waves: their share is printed against kernel cycles x 256 CUs ("LDS array busy"), the
"% of SQ_WAVE_CYCLES" of earlier summaries mixed the units."""
import csv
import glob
import re
import sys
from collections import defaultdict

XCDS, SIMDS, CUS = 8, 1024, 256
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
        q = m / 4.0 if c == "SQ_VALU_MFMA_BUSY_CYCLES" else m            # cycles -> quad-cycles
        extra = f"  {q / wcm:6.1%} of SQ_WAVE_CYCLES" if wcm and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" and not c.startswith("SQ_LDS_") else ""
        if c.startswith("SQ_LDS_") and "GRBM_GUI_ACTIVE" in cs:
            gui = cs["GRBM_GUI_ACTIVE"][0] / max(cs["GRBM_GUI_ACTIVE"][1], 1)
            if gui > 0:
                extra = f"  | LDS array busy {m / (gui / XCDS * CUS):6.1%} of kernel cycles x CUs"
        if c == "SQ_VALU_MFMA_BUSY_CYCLES" and "GRBM_GUI_ACTIVE" in cs:
            gui = cs["GRBM_GUI_ACTIVE"][0] / max(cs["GRBM_GUI_ACTIVE"][1], 1)
            if gui > 0:
                extra += f"  | per-SIMD pipe {m / (gui / XCDS * SIMDS):6.1%}"
        print(f"    {c:32s} {m:16.1f}{extra}")
