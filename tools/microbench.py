"""Kernel micro-benchmarks at the path's real shapes (run on the GPU box; writes JSON lines)."""
import json
import math
import sys
import time

import torch

sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main(out):
    res = []
    B = 64
    for dtype in (torch.bfloat16, torch.float32):
        for (M, N, K, tag) in ((B * 785, 1152, 384, "vit.qkv"), (B * 785, 384, 384, "vit.proj"), (B * 785, 1536, 384, "vit.fc1"),
                               (B * 785, 384, 1536, "vit.fc2"), (B * 385, 2048, 256, "dec.linear1"), (B * 385, 256, 2048, "dec.linear2"),
                               (8192, 8192, 8192, "square8k")):
            if dtype == torch.float32 and tag == "square8k":
                continue
            a = torch.randn(M, K, device="cuda").to(dtype)
            w = torch.randn(N, K, device="cuda").to(dtype)
            bias = torch.randn(N, device="cuda")
            o = torch.empty(M, N, device="cuda", dtype=dtype)
            t = timeit(lambda: h.gemm(a, w, bias=bias, out=o))
            t_ref = timeit(lambda: torch.addmm(bias.to(dtype), a, w.t()))
            res.append(dict(op="gemm", tag=tag, dtype=str(dtype), M=M, N=N, K=K, ms=t * 1e3, tflops=2 * M * N * K / t / 1e12,
                            hipblaslt_ms=t_ref * 1e3, hipblaslt_tflops=2 * M * N * K / t_ref / 1e12))
            print(res[-1], flush=True)
        for (H, Lq, Lk, hd, causal, tag) in ((6, 785, 785, 64, False, "vit.attn"), (8, 385, 385, 32, True, "dec.self"), (8, 385, 784, 32, False, "dec.cross")):
            Dm = H * hd
            q = torch.randn(B, Lq, Dm, device="cuda").to(dtype)
            k = torch.randn(B, Lk, Dm, device="cuda").to(dtype)
            v = torch.randn(B, Lk, Dm, device="cuda").to(dtype)
            t = timeit(lambda: h.attention(q, k, v, H, 1 / math.sqrt(hd), causal=causal))
            fl = 4 * B * H * Lq * Lk * hd * (0.5 if causal else 1.0)
            res.append(dict(op="attention", tag=tag, dtype=str(dtype), ms=t * 1e3, tflops=fl / t / 1e12))
            print(res[-1], flush=True)
        x = torch.randn(B * 785, 384, device="cuda")
        g = torch.ones(384, device="cuda")
        t = timeit(lambda: h.layernorm(x, g, g, 1e-6, out_dtype=dtype))
        byts = x.numel() * (4 + (2 if dtype == torch.bfloat16 else 4))
        res.append(dict(op="layernorm", dtype=str(dtype), ms=t * 1e3, gbps=byts / t / 1e9))
        print(res[-1], flush=True)
    with open(out, "w") as f:
        for r in res:
            f.write(json.dumps(r) + "\n")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/microbench.jsonl")
