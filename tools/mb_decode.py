"""Inference tail timing: encoder eval forward, KV-cached greedy decode (eager launches vs one replayed hipGraph per step), ScoreNet
scores + device Hungarian.  python tools/mb_decode.py [batch] [precision]"""
import os
import sys
import time

import torch

sys.path.insert(0, ".")
from pixelspointspolygons_amd import synthetic as O  # noqa: E402  (product-side synthetic inputs)
from pixelspointspolygons_amd.config import make_config  # noqa: E402
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
cfg = make_config("early_fusion_vit", precision=prec, device="cuda")
torch.manual_seed(42)
m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0).eval()
inp = O.make_inputs(B, seed=5)
d = {k: v.cuda() for k, v in inp.items()}


def timed(fn, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n, out


with torch.no_grad():
    lidar = (d["lidar_values"], d["lidar_offsets"])
    m.encoder(d["image"], lidar)
    t_enc, enc = timed(lambda: m.encoder(d["image"], lidar), 3)
    m.generate(enc, steps=8)
    t_eager, (tok_e, feats_e) = timed(lambda: m.generate(enc))
    t_first, _ = timed(lambda: m.generate(enc, graphs=True))
    t_capture, _ = timed(lambda: m.generate(enc, graphs=True))
    t_graph, (tok_g, feats_g) = timed(lambda: m.generate(enc, graphs=True), 3)
    assert os.environ.get("P3_MB_NOASSERT") or (torch.equal(tok_e, tok_g) and torch.equal(feats_e, feats_g))
    m.permutations(feats_g)
    t_perm, perm = timed(lambda: m.permutations(feats_g), 3)
print(f"B={B} {prec}: encoder {t_enc * 1e3:.1f} ms | decode 385 steps: eager launches {t_eager * 1e3:.0f} ms, graphs {t_graph * 1e3:.0f} ms "
      f"(capture pass {t_capture * 1e3:.0f} ms once) | scorenets + Hungarian {t_perm * 1e3:.1f} ms | "
      f"whole predict tail {B / (t_enc + t_graph + t_perm):.0f} tiles/s (eager decode: {B / (t_enc + t_eager + t_perm):.0f})")
