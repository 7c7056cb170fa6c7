"""Which python lines launch the ATen kernels of one train step (eager): torch.profiler with stacks, grouped by (op, innermost repo frame)."""
import sys, collections, torch
sys.path.insert(0, ".")
PREC = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"          # python tools/aten_sites.py [bf16 | fp32x3]
sys.argv = ["bench.py", "--lean", "--graph", "0"]
import bench
args = bench.parse()
from pixelspointspolygons_amd import synthetic as S
cfg, model, opt, reducer, pool, st = bench.build(args, "cuda:0", 0, PREC, S, 0, 1, False)
for i in range(3):
    st.step(pool[i % len(pool)])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    st.step(pool[0])
    torch.cuda.synchronize()
cnt = collections.Counter(); tim = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    t = getattr(ev, "self_device_time_total", 0)
    if t <= 0:
        continue
    frames = [f for f in (ev.stack or []) if "/repo/" in f and "bench.py" not in f]
    frame = frames[0].split("/repo/")[-1][:100] if frames else str(getattr(ev, "input_shapes", "?"))[:110]
    cnt[(ev.name, frame)] += 1; tim[(ev.name, frame)] += t
for key, n in cnt.most_common(70):
    print(f"{n:4d} {tim[key]:8.0f}us  {key[0]:24s} {key[1]}")
