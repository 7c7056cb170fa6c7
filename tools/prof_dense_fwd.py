"""The LiDAR stem's forward alone (no grad) at a given density, for `rocprofv3 --kernel-trace --stats -- python tools/prof_dense_fwd.py [points] [precision]`:
prints the event-timed ms per batch; the kernel stats of the run show where it goes."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from pixelspointspolygons_amd import synthetic as S  # noqa: E402
from pixelspointspolygons_amd.config import make_config  # noqa: E402
from pixelspointspolygons_amd.pointpillars import PointPillarsEncoder  # noqa: E402

npts = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
prec = sys.argv[2] if len(sys.argv) > 2 else "fp32x3"
dev = "cuda:0"
cfg = make_config("pointpillars_vit", precision=prec, device=dev)
stem = PointPillarsEncoder(cfg).to(dev).train()
inp = S.make_inputs(64, seed=777, n_points=npts, jitter=npts // 10)
vals, offs = inp["lidar_values"].to(dev), inp["lidar_offsets"].to(dev)
with torch.no_grad():
    for _ in range(3):
        stem((vals, offs), return_flattened=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        stem((vals, offs), return_flattened=True)
    e1.record()
    torch.cuda.synchronize()
print(f"{npts} points per tile, {prec}: {e0.elapsed_time(e1) / 10:.3f} ms per batch of 64 tiles")
