"""Times the fused FFL loss (value + both gradients, p3_ffl_loss) at bs x 224 x 224: python tools/mb_ffl_loss.py [batch]"""
import math
import sys
import time

import torch

sys.path.insert(0, ".")
from pixelspointspolygons_amd.config import make_config  # noqa: E402
from pixelspointspolygons_amd.ffl_losses import build_combined_loss  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H = 224
g = torch.Generator(device="cuda").manual_seed(0)
seg = torch.sigmoid(torch.randn(B, 1, H, H, device="cuda", generator=g)).requires_grad_(True)
cf = (2 * torch.tanh(torch.randn(B, 4, H, H, device="cuda", generator=g))).requires_grad_(True)
gt = (torch.rand(B, 3, H, H, device="cuda", generator=g) > 0.7).float()
angle = (torch.rand(B, 1, H, H, device="cuda", generator=g) * 2 - 1) * math.pi
crit = build_combined_loss(make_config("early_fusion_vit_cnn", model="ffl", device="cuda"))
pred, gtb = {"seg": seg, "crossfield": cf}, {"gt_polygons_image": gt, "gt_crossfield_angle": angle}


def step():
    total, _, _ = crit(pred, gtb, epoch=10.0)
    seg.grad = cf.grad = None
    total.backward()


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 20
for _ in range(n):
    step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
planes = 9 + 9 + 5 + 6 * 2 + 5 * 2            # pass 1 reads, pass 2 reads + grads + scratch, pass 3 scratch reads + grad read-modify-write
print(f"FFL loss value + gradients, B={B} {H}x{H}: {dt * 1e3:.3f} ms/step ({B / dt:.0f} tiles/s), ~{planes * B * H * H * 4 / dt / 1e12:.2f} TB/s of plane traffic")
