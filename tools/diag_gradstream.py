"""bf16 residual-gradient stream (P3_GRAD_BF16=1) against the fp32 one on the bench model: per-parameter gradient cosine and norm ratio of
one train-mode backward (same weights, same batch, dropout off), and the step time of both.   python tools/diag_gradstream.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
sys.argv = [sys.argv[0]]
import bench  # noqa: E402
from pixelspointspolygons_amd import ops, synthetic as S  # noqa: E402
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer  # noqa: E402
from pixelspointspolygons_amd.training import pix2poly_loss  # noqa: E402


def grads(flag, model, b):
    ops.GRAD_STREAM_BF16[0] = flag
    ops.clear_twins()
    for p in model.parameters():
        p.grad = None
    logits, perm = model(b.get("image"), (b["lidar_values"], b["lidar_offsets"]), b["y"][:, :-1])
    loss = pix2poly_loss(logits, perm, b["y"][:, 1:], b["y_perm"], 1.0, 10.0, 226)[0]
    loss.backward()
    torch.cuda.synchronize()
    return float(loss), {n: p.grad.float().clone() for n, p in model.named_parameters() if p.grad is not None}


def main():
    args = bench.parse()
    dev = "cuda:0"
    torch.cuda.set_device(0)
    args.batch = 16
    cfg = bench.make_cfg(args, dev, precision="bf16")
    torch.manual_seed(42)
    model = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0).train()
    model.decoder.set_dropout(0.0)
    b = bench.synth_batch(S, args, 0, 0, dev, "fusion")
    l0, g0 = grads(False, model, b)
    l1, g1 = grads(True, model, b)
    l2, g2 = grads(False, model, b)          # fp32 stream again: run-to-run noise floor (atomics order)
    worst, worst0 = [], []
    for n in g0:
        a, c, r = g0[n].flatten(), g1[n].flatten(), g2[n].flatten()
        if float(a.norm()) == 0:
            continue
        cos = float((a * c).sum() / (a.norm() * c.norm()).clamp_min(1e-30))
        cos0 = float((a * r).sum() / (a.norm() * r.norm()).clamp_min(1e-30))
        worst.append((1 - cos, float(c.norm() / a.norm()), n))
        worst0.append((1 - cos0, n))
    worst.sort(reverse=True)
    worst0.sort(reverse=True)
    enc = [w for w in worst if ".vit." in w[2] or "fusion" in w[2] or "image_embed" in w[2] or "lidar_embed" in w[2]]
    print(f"loss fp32-stream {l0:.6f} bf16-stream {l1:.6f}")
    print("worst 1-cos (bf16 gradient stream vs fp32 stream), norm ratio, parameter:")
    for w in worst[:8]:
        print(f"  {w[0]:.3e}  {w[1]:.4f}  {w[2]}")
    print(f"encoder parameters: max 1-cos {max(w[0] for w in enc):.3e}, median {sorted(w[0] for w in enc)[len(enc) // 2]:.3e} over {len(enc)} tensors")
    print(f"noise floor (fp32 stream twice): max 1-cos {worst0[0][0]:.3e} ({worst0[0][1]})")
    # step time, same process
    args.batch = 64
    for flag in (False, True, False, True):
        ops.reset_process_state()
        ops.GRAD_STREAM_BF16[0] = flag
        cfg, m, opt, red, pool, st = bench.build(args, dev, 0, "bf16", S, 0, 1, False)
        dt, loss = bench.timed_steps(st, pool, 20, 5, 1, dev)
        print(f"GRAD_STREAM_BF16={int(flag)}: {dt / 20 * 1e3:.3f} ms/step, loss {loss:.4f}", flush=True)
        opt.close()
        del st, m, opt, red, pool
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
