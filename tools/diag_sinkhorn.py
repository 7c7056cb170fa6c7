"""Diagnostic (GPU): per train step, the row spread of the Sinkhorn input (the linear-domain kernels need < 60) and the tile flags the
backward's first kernel left (1 = left to the log-domain kernel).   python tools/diag_sinkhorn.py [steps]"""
import sys

import torch

sys.path.insert(0, ".")
sys.argv = [sys.argv[0]] + sys.argv[1:]
import bench  # noqa: E402
from pixelspointspolygons_amd import hip, ops, synthetic as S  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    sys.argv = [sys.argv[0]]
    args = bench.parse()
    args.graph = int(__import__("os").environ.get("DIAG_GRAPH", "0"))
    dev = "cuda:0"
    torch.cuda.set_device(0)
    cfg, model, opt, reducer, pool, st = bench.build(args, dev, 0, "bf16", S, 0, 1, False)
    seen = {}
    orig = ops.sinkhorn_softmax

    def spy(scores, alpha, iters):
        if torch.cuda.is_current_stream_capturing():
            seen["spread"] = "graph"
            return orig(scores, alpha, iters)
        s = scores.detach()
        full = torch.cat([torch.cat([s, alpha.detach().float().expand(s.shape[0], s.shape[1], 1)], 2),
                          alpha.detach().float().expand(s.shape[0], 1, s.shape[2] + 1)], 1)
        spread = (full.max(2).values - full.min(2).values)
        seen["spread"] = (float(spread.max()), float(spread.mean()), float(s.abs().max()), float(alpha))
        return orig(scores, alpha, iters)
    ops.sinkhorn_softmax = spy
    import pixelspointspolygons_amd.pix2poly as P
    for i in range(steps):
        out = st.step(pool[i % len(pool)])
        torch.cuda.synchronize()
        ws = next((v for k, v in hip._ws_cache.items() if k[1] == "sinkhorn_bwd"), None)
        flags = ws[:64 * 4].view(torch.int32).cpu().tolist() if ws is not None else None
        print(f"step {i}: loss {float(out):.4f} spread max/mean {seen.get('spread')} wide tiles (bwd) {sum(flags) if flags else None}", flush=True)


if __name__ == "__main__":
    main()
