"""Stage timestamps of p3_decode_layer (last step of a 385-step decode, workgroup 0): needs a library built with -DDL_TIMING, e.g.
   tools/build_variant.sh tmp_ab/dl_timing.so decode_layer.hip -DDL_TIMING ; P3HIP_LIB=tmp_ab/dl_timing.so python tools/mb_decode_stages.py [batch]"""
import sys, torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import synthetic as O
from pixelspointspolygons_amd.config import make_config
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = make_config("early_fusion_vit", precision="bf16", device="cuda")
torch.manual_seed(42)
m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0).eval()
d = {k: v.cuda() for k, v in O.make_inputs(B, seed=5).items()}
with torch.no_grad():
    enc = m.encoder(d["image"], (d["lidar_values"], d["lidar_offsets"]))
    m.generate(enc, steps=385, graphs=True)
    m.generate(enc, steps=385, graphs=True)
    st = m.decoder._decode_state
    torch.cuda.synchronize()
    ts = st["dl_scratch"][0][B * 3 * 4 * 256:].view(torch.int64)[:14].cpu().tolist()
names = ["start", "x load", "in_proj", "cache write", "self attn", "so partial", "combine+LN1", "q proj", "cross attn", "co partial", "combine+LN2", "linear1", "linear2", "combine+LN3"]
for i in range(1, 14):
    print(f"{names[i]:14s} {(ts[i] - ts[i-1]) * 10:6d} ns")
print("total", (ts[13] - ts[0]) * 10, "ns")
