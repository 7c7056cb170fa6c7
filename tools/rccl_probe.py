"""RCCL availability probe for a 1-GPU box: python tools/rccl_probe.py
A 1-rank "nccl" process group is the most a single GPU allows (two ranks on one device are refused by RCCL).  It shows that
  (1) RCCL initialises on the box and an async all_reduce on a gradient-bucket-sized buffer completes;
  (2) whether an all_reduce can be CAPTURED into a hipGraph next to compute kernels and replayed (what a graph-safe reducer needs).
Prints one JSON line.  Run under `timeout`: a hang here must not take the box down."""
import json, os, sys, time
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29613")
res = {}
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
t0 = time.perf_counter()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
x = torch.ones(1 << 20, device=dev)
dist.all_reduce(x)
torch.cuda.synchronize()
res["init_s"] = round(time.perf_counter() - t0, 2)
res["eager_ok"] = bool((x == 1).all())
g = torch.full((22_000_000,), 2.0, device=dev)          # ~ one 88 MB fp32 gradient bucket
torch.cuda.synchronize(); t0 = time.perf_counter()
hs = [dist.all_reduce(g, async_op=True) for _ in range(10)]
for h in hs:
    h.wait()
torch.cuda.synchronize()
res["bucket_allreduce_us"] = round((time.perf_counter() - t0) / 10 * 1e6, 1)
# capture
try:
    y = torch.zeros(1 << 20, device=dev)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            y.add_(1.0); dist.all_reduce(y)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    y.zero_()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        y.add_(1.0)
        h = dist.all_reduce(y, async_op=True)
        z = y * 2.0                                  # independent compute issued while the collective is in flight
        h.wait()
        y.mul_(1.0)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    res["capture_ok"] = bool((y == 5.0).all())      # the capture pass does not execute; five replays add 1 each
    res["y0"] = float(y[0])
except Exception as e:                               # noqa: BLE001
    res["capture_ok"] = False
    res["capture_error"] = f"{type(e).__name__}: {str(e)[:300]}"
print(json.dumps(res), flush=True)
dist.destroy_process_group()
