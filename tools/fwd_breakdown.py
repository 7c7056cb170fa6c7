"""Forward-only loop for rocprofv3 --kernel-trace --stats (13 passes per process): python tools/fwd_breakdown.py [batch] [train|eval]"""
import sys

import torch

sys.path.insert(0, ".")
from pixelspointspolygons_amd import synthetic as O  # noqa: E402  (product-side synthetic inputs)
from pixelspointspolygons_amd import hip  # noqa: E402
from pixelspointspolygons_amd.config import make_config  # noqa: E402
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
mode = sys.argv[2] if len(sys.argv) > 2 else "train"
cfg = make_config("early_fusion_vit", precision="bf16", device="cuda", batch_size=B)
torch.manual_seed(42)
m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
m.train(mode == "train")
inp = {k: v.cuda() for k, v in O.make_inputs(B, seed=5).items()}
lidar = (inp["lidar_values"], inp["lidar_offsets"])
import time
with torch.no_grad():
    for _ in range(3):
        m(inp["image"], lidar, inp["y"][:, :-1])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        m(inp["image"], lidar, inp["y"][:, :-1])
    torch.cuda.synchronize()
print(f"forward ({mode} mode) B={B}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per pass (eager launches); 13 passes in this process")
