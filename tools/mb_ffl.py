"""FFL (early_fusion_vit_cnn) inference forward throughput on synthetic tiles (BASELINE configs[4] shape, forward only)."""
import sys, time, torch
sys.path.insert(0, ".")
from pixelspointspolygons_amd import synthetic as O  # noqa: E402  (product-side synthetic inputs)
from pixelspointspolygons_amd.config import make_config
from pixelspointspolygons_amd.ffl import FFLModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
cfg = make_config("early_fusion_vit_cnn", model="ffl", precision="bf16")
m = FFLModel(cfg, 0).eval()
d = O.make_inputs(B, seed=1)
img = d["image"].cuda()
nt = torch.nested.nested_tensor_from_jagged(d["lidar_values"].cuda(), d["lidar_offsets"].cuda())
with torch.no_grad():
    for _ in range(3):
        out = m({"image": img, "lidar": nt})
    torch.cuda.synchronize(); t0 = time.time()
    n = 10
    for _ in range(n):
        out = m({"image": img, "lidar": nt})
    torch.cuda.synchronize(); dt = (time.time() - t0) / n
gf = 2 * 129.5 * B
print(f"FFL early-fusion forward bs={B}: {dt*1e3:.2f} ms/batch, {B/dt:.1f} tiles/s, {gf/dt/1e3:.0f} TFLOP/s of the dense count ({gf/dt/1e3/2500:.1%} of 2.5 PF); peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")

# training step (forward + backward of a linear functional of the outputs; the FFL losses themselves live in the reference's trainer)
m.train()
g1 = torch.randn(B, 1, 224, 224, device="cuda"); g2 = torch.randn(B, 4, 224, 224, device="cuda")
def step():
    for p in m.parameters():
        p.grad = None
    out = m({"image": img, "lidar": nt})
    ((out["seg"] * g1).sum() + (out["crossfield"] * g2).sum()).backward()
for _ in range(2):
    step()
torch.cuda.synchronize(); t0 = time.time()
n = 5
for _ in range(n):
    step()
torch.cuda.synchronize(); dt = (time.time() - t0) / n
print(f"FFL early-fusion fwd+bwd bs={B}: {dt*1e3:.1f} ms/step, {B/dt:.1f} tiles/s, {3*gf/dt/1e3:.0f} TFLOP/s (3x forward count); peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
