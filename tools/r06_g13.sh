#!/bin/bash
# r06 g13: is the epilogue's cost its HBM traffic?  P3_AS_VAR=4 folds every row block's epilogue loads / stores onto rows 0..255 (1.5 MB per stream: cache-resident)
mkdir -p gpurun_out
O=gpurun_out/mb_as_13.txt
: > $O
for v in 0 4 2; do P3_AS_VAR=$v timeout 200 python tools/mb_as.py as >> $O 2>&1; done
python - >> $O 2>&1 <<'PY'
import torch, time
x = torch.empty(617 * 1024 * 1024 // 4, device="cuda")
for fn, name, nbytes in ((lambda: x.fill_(1.0), "fill 617 MB", x.numel() * 4), (lambda: x[: x.numel() // 2].copy_(x[x.numel() // 2:]), "copy 308 -> 308 MB", x.numel() * 4)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{name}: {ms * 1e3:.1f} us  {nbytes / ms / 1e9:.2f} TB/s")
PY
grep -v amdgpu.ids $O | tail -42
