"""Fill the @@...@@ placeholders of DESIGN.md's round-4 page from the committed artefacts (profiles/r04_bench.json, r04_train_step_graph_summary.txt)."""
import json, re, sys
d = json.loads(open("profiles/r04_bench.json").read().strip().splitlines()[-1])
ktot = re.search(r"total kernel time ([0-9.]+) ms", open("profiles/r04_train_step_graph_summary.txt").read()).group(1)
vals = {"STEP": f"{d['ms_per_step']:.1f}", "TPS": f"{d['value']:.0f}", "KTOT": ktot, "TRAFFIC": f"{d['roofline']['step_traffic_bytes'] / 1e9:.1f}",
        "ENCMS": f"{d['encoder_fwd']['ms_per_batch']:.2f}", "ENCFRAC": f"{d['encoder_fwd']['mfma_frac']:.3f}", "FP32TPS": f"{d['fp32_parity_mode']['value']:.0f}",
        "FFLTPS": f"{d['ffl']['value']:.0f}"}
s = open("DESIGN.md").read()
for k, v in vals.items():
    s = s.replace(f"@@{k}@@", v)
left = re.findall(r"@@[A-Z0-9]+@@", s)
open("DESIGN.md", "w").write(s)
print(vals, "left:", left)
