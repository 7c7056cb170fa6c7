#!/usr/bin/env python3
"""Headline benchmark of the Pix2Poly hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" = one reference train step (train/trainer_pix2poly.py:305-329) on one synthetic batch that is already resident
in HBM: forward (encoder + fusion + decoder + 2x ScoreNet + Sinkhorn) -> 1.0*CE + 10.0*BCE -> backward -> AdamW.
Metric (BASELINE.json): training tiles/s, whole job; `fwd_ms_per_tile` is reported in the same line.
Default workload = the configuration the metric is quoted on ("224px img + 3k-pt lidar"): early-fusion Pix2Poly
(ViT-S/8 + PointPillars stem, mnv = 64), 64 tiles per GPU, bf16 storage / fp32 accumulate.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# forward GFLOP per tile of the dense reference formulation (SURVEY §8d / BASELINE.md §2)
GFLOP_FWD = {"fusion_s8": 85.2, "image_s8": 80.9, "image_b16": 35.13 + 10.7 + 25.4, "lidar_s8": 80.9 - 0.116 + 0.149}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fusion_s8", choices=["fusion_s8", "image_s8", "image_b16", "lidar_s8"])
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--points", type=int, default=3000)
    ap.add_argument("--graph", type=int, default=1, help="capture the step in a hipGraph (single-GPU)")
    ap.add_argument("--sync-bn", type=int, default=1, help="N > 1: SyncBatchNorm like the reference's convert_sync_batchnorm (step runs eagerly)")
    ap.add_argument("--no-dropout", action="store_true", help="A/B only: decoder dropout off (the headline run keeps the reference's rates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fwd", action="store_true", help="skip the forward-only latency leg (profiling runs)")
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic batches cycled through")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the PCIe-inclusive leg (host uint8 tiles + point lists through the device input pipeline)")
    return ap.parse_args()


def make_cfg(args, dev):
    from pixelspointspolygons_amd.config import make_config
    enc = {"fusion_s8": "early_fusion_vit", "image_s8": "vit", "image_b16": "vit", "lidar_s8": "pointpillars_vit"}[args.workload]
    kw = {}
    if args.workload == "image_b16":
        kw = dict(patch_size=16, patch_feature_dim=768, vit_heads=12)
    cfg = make_config(enc, precision=args.precision, device=dev, batch_size=args.batch, **kw)
    if args.workload == "image_b16":
        cfg.experiment.encoder.type = cfg.experiment.encoder.vit.type = "vit_base_patch16_224.dino"
    return cfg


def synth_batch(S, args, rank, step, dev, kind):
    inp = S.make_inputs(args.batch, seed=1234 + 1000 * rank + step, n_points=args.points, jitter=args.points // 10)
    b = {"y": inp["y"].to(dev), "y_perm": inp["y_perm"].to(dev)}
    if kind != "lidar":
        b["image"] = inp["image"].to(dev)
    if kind != "image":
        b["lidar_values"], b["lidar_offsets"] = inp["lidar_values"].to(dev), inp["lidar_offsets"].to(dev)
    return b


class Stepper:
    """Static input buffers + (optionally) one captured hipGraph of forward + loss + backward + AdamW."""

    def __init__(self, model, opt, reducer, pool, kind, use_graph):
        from pixelspointspolygons_amd import ops
        from pixelspointspolygons_amd.training import pix2poly_loss
        self.ops = ops
        self.model, self.opt, self.reducer, self.kind = model, opt, reducer, kind
        self.loss_fn = pix2poly_loss
        cap = max(int(b["lidar_values"].shape[0]) for b in pool) if kind != "image" else 0
        self.static = {k: torch.empty_like(v) for k, v in pool[0].items() if k not in ("lidar_values",)}
        if kind != "image":
            self.static["lidar_values"] = torch.zeros((cap, 3), dtype=torch.float32, device=pool[0]["y"].device)
        self.graph = None
        self.eager_done = 0
        self.use_graph = use_graph
        self.out = None

    def load(self, b):
        for k, v in b.items():
            if k == "lidar_values":
                self.static[k][: v.shape[0]].copy_(v, non_blocking=True)
            else:
                self.static[k].copy_(v, non_blocking=True)

    def _fwd_bwd(self):
        s = self.static
        y = s["y"]
        self.ops.advance_rng(y.device)           # new decoder dropout masks every step (device counter: replays with the graph)
        lidar = (s["lidar_values"], s["lidar_offsets"]) if self.kind != "image" else None
        logits, perm = self.model(s.get("image"), lidar, y[:, :-1])
        loss, ce, bce = self.loss_fn(logits, perm, y[:, 1:], s["y_perm"], 1.0, 10.0, 226)
        self.opt.zero_grad()
        loss.backward()
        return loss.detach()

    def step(self, b):
        self.load(b)
        self.opt.prepare_step()
        if self.use_graph and self.eager_done >= 2:
            single = self.reducer.world == 1
            if self.graph is None:
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = self._fwd_bwd()
                    if single:
                        self.opt.apply(1.0)
            self.graph.replay()
            if not single:                       # N > 1: graph = forward + backward; bucketed RCCL all-reduce + AdamW stay eager
                self.opt.apply(self.reducer.finish())
        else:
            self.out = self._fwd_bwd()
            self.opt.apply(self.reducer.finish())
            self.eager_done += 1
        return self.out

    def forward_only(self, b, use_graph=True):
        """train-mode forward (batch statistics) without autograd; captured in its own hipGraph after two eager passes."""
        self.load(b)
        s = self.static
        lidar = (s["lidar_values"], s["lidar_offsets"]) if self.kind != "image" else None

        def run():
            with torch.no_grad():
                return self.model(s.get("image"), lidar, s["y"][:, :-1])
        if not (use_graph and self.use_graph):
            return run()
        self.fwd_eager = getattr(self, "fwd_eager", 0)
        if self.fwd_eager < 2:
            self.fwd_eager += 1
            return run()
        if getattr(self, "fwd_graph", None) is None:
            torch.cuda.synchronize()
            self.fwd_graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd_graph):
                self.fwd_out = run()
        self.fwd_graph.replay()
        return self.fwd_out


def host_feed_leg(S, args, st, dev, kind, rank):
    """PCIe-inclusive rate (never `value`): the same train step fed from HOST memory through pixelspointspolygons_amd.input_pipeline -
    uint8 HWC tiles + untransformed point lists + a D4 element per tile packed into pinned staging, H2D + D4/Normalize kernels on a
    copy stream overlapped with the previous step."""
    import numpy as np
    from pixelspointspolygons_amd.input_pipeline import DevicePrefetcher
    host_pool = []
    for s_ in range(args.pool):
        inp = S.make_inputs(args.batch, seed=1234 + 1000 * rank + s_, n_points=args.points, jitter=args.points // 10)
        hb = {"y": inp["y"], "y_perm": inp["y_perm"], "group": np.random.default_rng(s_).integers(0, 8, size=args.batch)}
        if kind != "lidar":
            hb["image"] = (inp["image"].permute(0, 2, 3, 1) * 255.0).round().to(torch.uint8).contiguous()
        if kind != "image":
            off = inp["lidar_offsets"].tolist()
            hb["lidar"] = [inp["lidar_values"][off[b]:off[b + 1]].numpy() for b in range(args.batch)]
        host_pool.append(hb)
    n_warm, n = 3, max(5, min(args.steps, 20))
    pf = DevicePrefetcher((host_pool[i % len(host_pool)] for i in range(n_warm + n)), dev, max_points=int(args.batch * args.points * 1.5))
    t0 = None
    for i, b in enumerate(pf):
        if i == n_warm:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        st.step(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    img_mb = args.batch * 224 * 224 * 3 / 1e6 if kind != "lidar" else 0.0
    return {"value": round(args.batch * n / dt, 2), "unit": "tiles/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n,
            "host_bytes_per_step_mb": round(img_mb + (args.batch * args.points * 12 / 1e6 if kind != "image" else 0.0) + args.batch * (386 * 8 + 192 * 192 * 4) / 1e6, 1),
            "what": "uint8 HWC tiles + jagged points + tokens from pinned host memory, D4 + Normalize + HWC->CHW on the device, double buffered"}


def cpu_baseline(args, kind):
    """The oracle (CPU restatement, kind = "port") timed on this box's host cores on a bounded sample of the same workload.
    The only place bench.py touches oracle/."""
    from oracle import p3_oracle as O
    import torch.nn.functional as F  # noqa: F401
    B = 2
    cfgv = O.VIT_S8 if args.workload != "image_b16" else O.VIT_B16
    sd = O.make_state_dict({"fusion_s8": "fusion", "image_s8": "image", "image_b16": "image", "lidar_s8": "lidar"}[args.workload], cfgv, seed=42)
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    params = [v for v in p.values() if v.is_floating_point() and v.requires_grad]
    opt = torch.optim.AdamW(params, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))
    inp = O.make_inputs(B, seed=1234, n_points=args.points, jitter=args.points // 10)
    img = inp["image"] if kind != "lidar" else None
    lidar = (inp["lidar_values"], inp["lidar_offsets"]) if kind != "image" else None

    def one():
        logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], img, lidar, cfg=cfgv, training=True)
        loss, _, _ = O.pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    one()
    t0 = time.time()
    n = 0
    while n < 2 or (time.time() - t0 < 12 and n < 6):
        one()
        n += 1
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 4), "unit": "tiles/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle train step (fwd+CE+10*BCE+bwd+AdamW, fp32), batch {B}, {n} timed steps after 1 warm-up, host CPU"}


def pmc_traffic(kernel_label):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate
    runs of `bench.py --graph 0`; FETCH doubled per the gfx950 note in MI355X_MICROARCH.md, WRITE as reported; tools/pmc_traffic.py).
    bench.py cannot collect counters on itself, so this is the number of the same kernel on the same workload from profiles/."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if not os.path.exists(path) or not kernel_label.startswith("gemm_kernel<bf16,plain"):
        return None, None
    with open(path) as fh:
        t = json.load(fh)
    tot = n = 0.0
    for k, v in t.items():
        parts = [x.strip() for x in k[k.find("<") + 1:k.rfind(">")].split(",")] if "<" in k else []
        if k.startswith("gemm_kernel<unsigned short,") and len(parts) >= 3 and parts[2] == "0":   # bf16 in, bf16 / f32 out, plain A
            tot += (v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"]
            n += v["launches"]
    return (round(tot / n) if n else None), "profiles/r01_pmc_traffic.json (FETCH_SIZE x2 + WRITE_SIZE, mean per launch over the train step)"


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for the 1-GPU development boxes (the N > 1 control flow is exercised with two ranks sharing device 0 over gloo):
    # P3_BENCH_BACKEND=gloo P3_BENCH_ONE_DEVICE=1.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL.
    if os.environ.get("P3_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("P3_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, init_method="env://")
    from pixelspointspolygons_amd import synthetic as S        # synthetic-input generator (oracle/ is imported by the cpu_baseline leg only)
    from pixelspointspolygons_amd import hip
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, GradBucketReducer
    from pixelspointspolygons_amd.vision_transformer import compute_dtype

    kind = {"fusion_s8": "fusion", "image_s8": "image", "image_b16": "image", "lidar_s8": "lidar"}[args.workload]
    cfg = make_cfg(args, dev)
    torch.manual_seed(42)                    # reference seed (train/trainer.py:214); random-init weights (no checkpoints offline)
    tk = Tokenizer(cfg)
    model = Pix2PolyModel(cfg, tk.vocab_size, local)
    model.train()
    if args.no_dropout:
        model.decoder.set_dropout(0.0)
    # decoder dropout stays at the reference's training defaults (0.1 in nn.TransformerDecoderLayer incl. the attention
    # probabilities, 0.05 on both positional sums, model_pix2poly.py:136-143): fused into the GEMM epilogues / attention kernels
    # N > 1 with SyncBatchNorm (the reference's DDP setup, model_pix2poly.py:326-328): the 18 small statistic all-reduces sit inside
    # forward/backward, so the step runs eagerly (measured on one GPU: eager == graph within 0.3 %, the step is GPU-bound); gradients
    # go straight into the flat arena and are all-reduced bucket by bucket after backward.
    sync_bn = world > 1 and bool(args.sync_bn)
    if sync_bn:
        from pixelspointspolygons_amd import ops
        ops.SYNC_BN[0] = True
        args.graph = 0
    opt = FlatAdamW(model, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), compute_dtype=compute_dtype(cfg),
                    direct_grad=bool(args.graph) or world == 1 or sync_bn)   # hook-driven overlap needs autograd's AccumulateGrad
    opt.set_linear_schedule(200 * 1000)
    reducer = GradBucketReducer(opt, overlap=not args.graph and not sync_bn)   # hooks (overlap) only on the plain eager path
    pool = [synth_batch(S, args, rank, s, dev, kind) for s in range(args.pool)]
    st = Stepper(model, opt, reducer, pool, kind, bool(args.graph))

    for i in range(max(args.warmup, 3 if args.graph else 0)):   # 2 eager steps + the capture step stay untimed
        st.step(pool[i % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = st.step(pool[i % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    loss_val = float(out)

    # forward-only latency (eval of the same model state, no grad)
    fwd_ms = float("nan")
    if not args.no_fwd:
        for i in range(4):
            st.forward_only(pool[0])
        torch.cuda.synchronize()
        nf = max(3, min(args.steps, 10))
        t1 = time.perf_counter()
        for i in range(nf):
            st.forward_only(pool[i % len(pool)])
        torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t1) / nf * 1e3
    # inference forward: model.eval() (running BatchNorm statistics, no dropout), eager launches, same batch shape
    fwd_eval_ms = float("nan")
    if not args.no_fwd:
        model.eval()
        for i in range(3):
            st.forward_only(pool[0], use_graph=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(5):
            st.forward_only(pool[i % len(pool)], use_graph=False)
        torch.cuda.synchronize()
        fwd_eval_ms = (time.perf_counter() - t1) / 5 * 1e3
        model.train()

    feed = None
    if world == 1 and not args.no_host_feed and not args.no_fwd:
        feed = host_feed_leg(S, args, st, dev, kind, rank)

    # dominant-kernel timing with HIP events on the launch stream (instrumented steps, after the timed region)
    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        from pixelspointspolygons_amd import ops as _ops
        was_sync, _ops.SYNC_BN[0] = _ops.SYNC_BN[0], False     # rank-0-only leg: no collectives here
        hip.KTIMER.enable()
        for i in range(3):
            st.forward_only(pool[i % len(pool)], use_graph=False)
        torch.cuda.synchronize()
        kt = hip.KTIMER.summary()
        hip.KTIMER.disable()
        _ops.SYNC_BN[0] = was_sync
        if kt:
            name, rec = max(kt.items(), key=lambda kv: kv[1]["ms"])
            peak_tf = 2500.0 if args.precision == "bf16" else 157.3
            peak_gb = 8000.0
            sec = rec["ms"] * 1e-3
            ach_tf = rec["flop"] / sec / 1e12
            ach_gb = rec["bytes"] / sec / 1e9 if rec["bytes"] else 0.0
            ai = rec["flop"] / rec["bytes"] if rec["bytes"] else float("inf")
            ridge = peak_tf * 1e12 / (peak_gb * 1e9)
            traffic, tsrc = pmc_traffic(name)
            common = {"kernel": name, "launches": rec["n"], "avg_launch_us": round(rec["ms"] * 1e3 / rec["n"], 2),
                      "share_of_fwd_kernel_time": round(rec["ms"] / sum(r["ms"] for r in kt.values()), 3),
                      "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
                      "traffic": traffic, "traffic_source": tsrc,
                      "algorithmic_bytes_per_launch": round(rec["bytes"] / rec["n"]), "algorithmic_flop_per_launch": round(rec["flop"] / rec["n"])}
            # the bound is the one the launch mix sits under: K = 384 GEMMs with fp32 residual / bf16 outputs move 77-290 FLOP per
            # byte, below the 312 FLOP/B ridge of MI355X (2.5 PF / 8 TB/s) -> HBM bound; both fractions are reported
            if ai < ridge:
                roofline = {"bound": "hbm", "achieved": round(ach_gb, 1), "peak": peak_gb, "unit": "GB/s", "frac": round(ach_gb / peak_gb, 4),
                            "mfma_achieved_tflops": round(ach_tf, 2), "mfma_frac": round(ach_tf / peak_tf, 4), **common}
            else:
                roofline = {"bound": "mfma", "achieved": round(ach_tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach_tf / peak_tf, 4),
                            "hbm_achieved_gbs": round(ach_gb, 1), "hbm_frac": round(ach_gb / peak_gb, 4), **common}

    if rank == 0:
        tiles = args.batch * world * args.steps
        line = {
            "metric": "training tiles/sec (224px img + 3k-pt lidar)", "value": round(tiles / dt, 2), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": f"pix2poly_{args.workload}_bs{args.batch}x{world}", "tiles_per_gpu": args.batch, "points_per_tile": args.points,
                       "hip_graph": bool(args.graph), "sync_bn": sync_bn, "decoder_dropout": "off (A/B run)" if args.no_dropout else "reference defaults (0.1 / 0.05)",
                       "step": "fwd+CE+10*BCE+bwd+AdamW", "parallelism": f"dp{world}"},
            "fwd_ms_per_tile": round(fwd_ms / args.batch, 4), "fwd_ms_per_batch": round(fwd_ms, 3),
            "fwd_eval_ms_per_tile": round(fwd_eval_ms / args.batch, 4),
            "fwd_mfma_frac_of_2.5PF": round(GFLOP_FWD[args.workload] * args.batch / (fwd_ms * 1e-3) / 1e3 / 2500.0, 4) if args.precision == "bf16" else None,
            "final_loss": round(loss_val, 4),
            "roofline": roofline,
        }
        if feed is not None:
            line["pcie_inclusive"] = feed
        if not args.no_cpu_baseline and world == 1:       # reported at N = 1 only (the host cores are shared by the ranks otherwise)
            line["cpu_baseline"] = cpu_baseline(args, kind)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                                    # rank 0 runs the instrumented leg alone: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
