#!/usr/bin/env python3
"""Headline benchmark of the Pix2Poly / FFL hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py starts `python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 bench.py <same arguments>` itself as a CHILD process - before anything here has touched the
GPU - relays its output and exits with its code (the reference's launch is `torchrun --nproc_per_node=N scripts/train.py`, README.md:540,
and `setup_ddp`, misc/shared_utils.py:205-230).  Under a launcher (the driver's torch.distributed.run) the ranks check WORLD_SIZE == N.

A "step" = one reference train step (train/trainer_pix2poly.py:305-329) on one synthetic batch that is already resident
in HBM: forward (encoder + fusion + decoder + 2x ScoreNet + Sinkhorn) -> 1.0*CE + 10.0*BCE -> backward -> AdamW.
Metric (BASELINE.json): training tiles/s, whole job; `fwd_ms_per_tile` is reported in the same line.
Default workload = the configuration the metric is quoted on ("224px img + 3k-pt lidar"): early-fusion Pix2Poly
(ViT-S/8 + PointPillars stem, mnv = 64), 64 tiles per GPU, in the precision that meets north_star's tolerance at the highest rate:
`--precision fp32x3` (fp32 storage, every product as bf16 x 3 on the bf16 MFMA; logits within 1e-3 of the oracle and token indices bit-exact - the line carries
its own measured error as `fp32x3_vs_oracle`).  `bf16_mode` (bf16 storage, its measured error beside it) and `fp32_exact_mode` are sub-objects of the same line.
`--workload ffl_fusion` = BASELINE configs[4]: FFL early_fusion_vit_cnn, step = forward + FFL criterion + backward + AdamW.

Extra objects in the JSON line (N = 1): `roofline` (dominant kernel, HIP events on the launch stream; step-level traffic from the
committed PMC passes), `encoder_fwd` (the fused ViT + LiDAR encoder forward alone: the quantity the north star's MFMA fraction is defined on),
`fp32_exact_mode` (the same step on the exact fp32 MFMA path) and `bf16_mode` (bf16 storage - faster, outside the tolerance),
`bf16_vs_oracle` / `fp32_vs_oracle` / `fp32x3_vs_oracle` (MEASURED error of each bench model on two tiles of the
bench batch against the oracle on the host), `dense_lidar` (the LiDAR stem at 3 k and 40 k points per tile), `ffl` (BASELINE configs[4], 5 captured steps), `image_b16` (BASELINE configs[1]: the ViT-B/16 image-only step at bs 64 in bf16 and fp32x3),
`predict` (BASELINE configs[0]: image-only model, batch 1, 385-step greedy decode of the demo tile), `pcie_inclusive` (host-fed step, never `value`),
`cpu_baseline` (the oracle on this box's host cores), `rank_ms_per_step` (min / max over the ranks).
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Forward GFLOP per tile that the kernels EXECUTE: dense reference count minus the ScoreNet layer-1 work that the
# separable formulation skips (2 x ScoreNet: 12.688 -> 3.074 GMAC).  The dense reference counts ride along for comparison only.
_SCORENET_SKIPPED = 2.0 * (12.688 - 3.074)
GFLOP_FWD_DENSE = {"fusion_s8": 85.2, "image_s8": 80.9, "image_b16": 35.13 + 10.7 + 25.4, "lidar_s8": 80.9 - 0.116 + 0.149, "ffl_fusion": 259.0}
GFLOP_FWD = {k: (v if k == "ffl_fusion" else round(v - _SCORENET_SKIPPED, 2)) for k, v in GFLOP_FWD_DENSE.items()}
# "fused ViT + LiDAR forward" alone (stem + fusion conv + 12 blocks + pool), the quantity north_star's >= 40 % target is defined on
GFLOP_ENC = {"fusion_s8": 49.1, "image_s8": 2 * 22.405, "image_b16": 35.13, "lidar_s8": 2 * (22.347 + 0.0745)}
# algorithmic HBM bytes of one train step at 64 tiles: 1.14 MB/tile of inputs + outputs, AdamW 16 B per parameter
STEP_ALGO_BYTES = {"fusion_s8": 64 * 1.14e6 + 16 * 34.6e6}
PMC_FILES = {"fp32x3": ("r06_pmc_traffic_fp32x3.json", "r05_pmc_traffic_fp32x3.json"), "bf16": ("r05_pmc_traffic_bf16.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"), "fp32": ()}
# MFMA peak a mode is priced against: bf16 dense 2.5 PF; 'fp32x3' issues three bf16 MFMAs per algorithmic product -> 2.5 PF / 3; exact fp32 MFMA 157.3 TF
PEAK_TF = {"bf16": 2500.0, "fp32x3": 2500.0 / 3.0, "fp32": 157.3}
DTYPE_NAME = {"bf16": "bf16", "fp32x3": "f32 (products as bf16x3 on the bf16 MFMA, fp32 accumulate)", "fp32": "f32"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="fusion_s8", choices=["fusion_s8", "image_s8", "image_b16", "lidar_s8", "ffl_fusion"])
    ap.add_argument("--batch", type=int, default=64, help="tiles per GPU")
    ap.add_argument("--precision", default="fp32x3", choices=["bf16", "fp32", "fp32x3"],
                    help="fp32x3 (default): the mode that meets north_star's 1e-3 / bit-exact-index tolerance at the highest rate; bf16: faster, 1e-2; fp32: exact fp32 MFMA")
    ap.add_argument("--points", type=int, default=3000)
    ap.add_argument("--graph", type=int, default=1, help="capture the step in a hipGraph")
    ap.add_argument("--sync-bn", type=int, default=1, help="N > 1: SyncBatchNorm like the reference's convert_sync_batchnorm")
    ap.add_argument("--no-dropout", action="store_true", help="A/B only: decoder dropout off (the headline run keeps the reference's rates)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-fwd", action="store_true", help="skip the forward-only latency legs (profiling runs)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="skip the timed legs of the other two precisions (bf16_mode, fp32_exact_mode)")
    ap.add_argument("--no-ffl", action="store_true", help="skip the short FFL (configs[4]) sub-run of the default line")
    ap.add_argument("--no-b16", action="store_true", help="skip the ViT-B/16 image-only (configs[1]) sub-run of the default line")
    ap.add_argument("--no-predict", action="store_true", help="skip the configs[0] predict leg (image-only, batch 1, 385-step decode)")
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic batches cycled through")
    ap.add_argument("--no-host-feed", action="store_true", help="skip the PCIe-inclusive leg (host uint8 tiles + point lists through the device input pipeline)")
    ap.add_argument("--lean", action="store_true", help="timed loop only (= all --no-* switches; profiling / A-B runs)")
    a = ap.parse_args()
    if a.lean:
        a.no_cpu_baseline = a.no_kernel_timing = a.no_fwd = a.no_fp32_leg = a.no_predict = a.no_host_feed = a.no_ffl = a.no_b16 = True
    return a


KIND = {"fusion_s8": "fusion", "image_s8": "image", "image_b16": "image", "lidar_s8": "lidar", "ffl_fusion": "fusion"}


def make_cfg(args, dev, precision=None, workload=None, batch=None):
    from pixelspointspolygons_amd.config import make_config
    wl = workload or args.workload
    prec = precision or args.precision
    if wl == "ffl_fusion":
        return make_config("early_fusion_vit_cnn", model="ffl", precision=prec, device=dev, batch_size=batch or args.batch)
    enc = {"fusion_s8": "early_fusion_vit", "image_s8": "vit", "image_b16": "vit", "lidar_s8": "pointpillars_vit"}[wl]
    kw = {}
    if wl == "image_b16":
        kw = dict(patch_size=16, patch_feature_dim=768, vit_heads=12)
    cfg = make_config(enc, precision=prec, device=dev, batch_size=batch or args.batch, **kw)
    if wl == "image_b16":
        cfg.experiment.encoder.type = cfg.experiment.encoder.vit.type = "vit_base_patch16_224.dino"
    return cfg


def synth_batch(S, args, rank, step, dev, kind, ffl=False):
    inp = S.make_inputs(args.batch, seed=1234 + 1000 * rank + step, n_points=args.points, jitter=args.points // 10)
    b = {}
    if ffl:
        b.update({k: v.to(dev) for k, v in S.make_ffl_targets(args.batch, seed=4321 + 1000 * rank + step).items()})
    else:
        b.update({"y": inp["y"].to(dev), "y_perm": inp["y_perm"].to(dev)})
    if kind != "lidar":
        b["image"] = inp["image"].to(dev)
    if kind != "image":
        b["lidar_values"], b["lidar_offsets"] = inp["lidar_values"].to(dev), inp["lidar_offsets"].to(dev)
    return b


class Stepper:
    """Static input buffers + (optionally) one captured hipGraph of forward + loss + backward (+ bucketed all-reduce launches) + AdamW."""

    def __init__(self, model, opt, reducer, pool, kind, use_graph, criterion=None):
        from pixelspointspolygons_amd import ops
        from pixelspointspolygons_amd.training import pix2poly_loss
        self.ops = ops
        self.model, self.opt, self.reducer, self.kind = model, opt, reducer, kind
        self.loss_fn, self.criterion = pix2poly_loss, criterion
        cap = max(int(b["lidar_values"].shape[0]) for b in pool) if kind != "image" else 0
        self.static = {k: torch.empty_like(v) for k, v in pool[0].items() if k not in ("lidar_values",)}
        if kind != "image":
            self.static["lidar_values"] = torch.zeros((cap, 3), dtype=torch.float32, device=next(iter(pool[0].values())).device)
        self.graph = None
        self.eager_done = 0
        # No hipGraph capture in a process that issues collectives: the reducer's all-reduces are not capturable (graph_safe False), and
        # ProcessGroupNCCL's watchdog thread polls its events while ANY stream captures -> hipErrorStreamCaptureUnsupported aborts
        # the process (measured with a 1-rank RCCL group, tools/rccl_probe.py).  N > 1 steps and forward legs run as plain launches;
        # at N = 1 graph and eager steps time the same (48.66 vs 48.70 ms, profiles/r02_graph_vs_eager.txt).
        self.use_graph = use_graph and (not reducer.active or reducer.graph_safe)
        self.out = None

    def load(self, b):
        for k, v in b.items():
            if k not in self.static:
                continue
            if k == "lidar_values":
                self.static[k][: v.shape[0]].copy_(v, non_blocking=True)
            else:
                self.static[k].copy_(v, non_blocking=True)

    def _lidar(self):
        s = self.static
        return (s["lidar_values"], s["lidar_offsets"]) if self.kind != "image" else None

    def _forward(self):
        s = self.static
        if self.criterion is not None:       # FFL: dict batch in, {"seg", "crossfield"} out (model_ffl.py:99-104)
            lidar = self._lidar()
            nt = torch.nested.nested_tensor_from_jagged(lidar[0], lidar[1]) if lidar is not None else None
            return self.model({"image": s.get("image"), "lidar": nt})
        return self.model(s.get("image"), self._lidar(), s["y"][:, :-1])

    def _fwd_bwd(self):
        s = self.static
        dev = next(iter(s.values())).device
        self.opt.zero_grad()
        self.ops.advance_rng(dev)               # new decoder dropout masks every step (device counter: replays with the graph)
        out = self._forward()
        if self.criterion is not None:
            loss = self.criterion(out, {"gt_polygons_image": s["gt_polygons_image"], "gt_crossfield_angle": s["gt_crossfield_angle"]},
                                  normalize=True, epoch=10.0)[0]
        else:
            loss = self.loss_fn(out[0], out[1], s["y"][:, 1:], s["y_perm"], 1.0, 10.0, 226)[0]
        loss.backward()
        return loss.detach()

    def step(self, b):
        self.load(b)
        self.opt.prepare_step()
        if self.use_graph and self.eager_done >= 2:
            if self.graph is None:
                torch.cuda.synchronize()
                self.graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph):
                    self.out = self._fwd_bwd()
                    self.opt.apply(self.reducer.finish())
            self.graph.replay()
        else:
            self.out = self._fwd_bwd()
            self.opt.apply(self.reducer.finish())
            self.eager_done += 1
        return self.out

    def step_eager(self, b):
        """one train step with plain launches (roofline leg: HIP events around each launch cannot be recorded inside a graph replay)"""
        self.load(b)
        self.opt.prepare_step()
        self.out = self._fwd_bwd()
        self.opt.apply(self.reducer.finish())
        return self.out

    def forward_only(self, b, use_graph=True):
        """train-mode forward (batch statistics) without autograd; captured in its own hipGraph after two eager passes."""
        self.load(b)

        def run():
            with torch.no_grad():
                return self._forward()
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


def timed_steps(st, pool, steps, warmup, world, dev):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; max over ranks."""
    world = 2 if (dist.is_available() and dist.is_initialized()) else 1     # a process group (also the forced 1-rank one) => collective bracket
    for i in range(warmup):
        st.step(pool[i % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = None
    for i in range(steps):
        out = st.step(pool[i % len(pool)])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    spread = None
    if world > 1:
        t = torch.tensor([dt, -dt], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)          # [max over ranks, -min over ranks]
        dt = float(t[0])
        spread = (-float(t[1]), float(t[0]))
    timed_steps.rank_spread = spread
    return dt, float(out)


def host_feed_leg(S, args, st, dev, kind, rank):
    """PCIe-inclusive rate (never `value`): the same train step fed from HOST memory through pixelspointspolygons_amd.input_pipeline -
    uint8 HWC tiles + untransformed point lists + a D4 element per tile packed into pinned staging by a feeder thread, H2D + D4 /
    Normalize kernels on a copy stream, persistent device staging (no allocation inside the loop).  Median of 3 repeats."""
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
    n_warm, n, reps = 6, max(5, min(args.steps, 20)), 3
    pf = DevicePrefetcher((host_pool[i % len(host_pool)] for i in range(n_warm + reps * n)), dev, max_points=int(args.batch * args.points * 1.5))
    times, t0 = [], None
    for i, b in enumerate(pf):
        if i >= n_warm and (i - n_warm) % n == 0:
            torch.cuda.synchronize()
            now = time.perf_counter()
            if t0 is not None:
                times.append(now - t0)
            t0 = now
        st.step(b)
    torch.cuda.synchronize()
    times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    img_mb = args.batch * 224 * 224 * 3 / 1e6 if kind != "lidar" else 0.0
    return {"value": round(args.batch * n / dt, 2), "unit": "tiles/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n, "repeats": len(times),
            "ms_per_step_all": [round(t / n * 1e3, 3) for t in times], "min_ms_per_step": round(min(times) / n * 1e3, 3),     # host-side noise (other tenants on the box's cores) shows as spread between the repeats
            "host_bytes_per_step_mb": round(img_mb + (args.batch * args.points * 12 / 1e6 if kind != "image" else 0.0) + args.batch * (386 * 8 + 192 * 192 * 4) / 1e6, 1),
            "what": "uint8 HWC tiles + jagged points + tokens from pinned host memory (packed by a feeder thread), D4 + Normalize + HWC->CHW on the "
                    "device, persistent triple-buffered staging; median of the repeats"}


def _stats(samples):
    return {"min": round(min(samples), 4), "median": round(statistics.median(samples), 4), "n": len(samples)}


def cpu_baseline(args, kind, probes=None):
    """The oracle (CPU restatement, kind = "port") timed on this box's host cores on a bounded sample of the same workload
    (SURVEY §8d protocol, bounded: forward at B in {1, 8} (>= 3 timed, min and median ms/tile), train step at B = 4 (>= 5 timed
    iterations, tiles/s = median), s/tile of the literal 385-step greedy decode).  The legs stop at a wall-clock budget once their minimum
    iteration count is reached; the number of iterations that ran is reported.  The only place bench.py touches oracle/."""
    from oracle import p3_oracle as O
    cfgv = O.VIT_S8 if args.workload != "image_b16" else O.VIT_B16
    okind = {"fusion": "fusion", "image": "image", "lidar": "lidar"}[kind]
    # the checker's other job: `<dtype>_vs_oracle` = the MEASURED error of the bench's own model (eval mode, the first tiles of the bench batch,
    # parity_probe) against the oracle run on the host with the same weights and inputs - so the headline states its own error
    vs = {}
    for name, pr in (probes or {}).items():
        with torch.no_grad():
            rl, rp = O.pix2poly_forward({k: v.clone() for k, v in pr["sd"].items()}, pr["y"], pr["image"], pr["lidar"],
                                        cfg=O.VIT_B16 if name.startswith("image_b16_") else cfgv, training=False)
        rel = lambda a, b: float((a.double() - b.double()).abs().max() / b.double().abs().max())
        # element-wise: every value against ITS OWN size, floored at 1 % of the largest reference value (the norm-wise figure alone hides small entries)
        erel = lambda a, b: float(((a.double() - b.double()).abs() / b.double().abs().clamp_min(1e-2 * float(b.double().abs().max()))).max())
        am = (pr["logits"].argmax(-1) == rl.argmax(-1))
        vs[name] = {"tiles": int(pr["y"].shape[0]), "logits_rel": float(f"{rel(pr['logits'], rl):.3e}"), "perm_rel": float(f"{rel(pr['perm'], rp):.3e}"),
                    "logits_rel_elementwise": float(f"{erel(pr['logits'], rl):.3e}"), "perm_rel_elementwise": float(f"{erel(pr['perm'], rp):.3e}"),
                    "argmax_agree": round(float(am.float().mean()), 6), "argmax_positions": int(am.numel()),
                    "what": "eval-mode forward of the bench model on tiles 0..n-1 of the bench batch vs oracle.pix2poly_forward on the host; *_rel = max-abs error / max-abs "
                            "reference, *_rel_elementwise = max over elements of |a - b| / max(|b|, 1e-2 max|b|)"}
    sd = O.make_state_dict(okind, cfgv, seed=42)

    def inputs(B):
        inp = O.make_inputs(B, seed=1234, n_points=args.points, jitter=args.points // 10)
        img = inp["image"] if kind != "lidar" else None
        lidar = (inp["lidar_values"], inp["lidar_offsets"]) if kind != "image" else None
        return inp, img, lidar

    def run(fn, warm, max_n, budget, min_n=2):
        for _ in range(warm):
            fn()
        ts, t_end = [], time.time() + budget
        while len(ts) < max_n and (len(ts) < min_n or time.time() < t_end):
            t0 = time.time()
            fn()
            ts.append(time.time() - t0)
        return ts

    fwd, train = {}, {}
    for B in (1, 8):
        inp, img, lidar = inputs(B)
        with torch.no_grad():
            ts = run(lambda: O.pix2poly_forward(sd, inp["y"][:, :-1], img, lidar, cfg=cfgv, training=False), 3 if B == 1 else 1, 10, 10.0 if B == 1 else 40.0, min_n=3)      # 10 timed iterations per leg (B = 8: ~2.5 s each)
        fwd[f"B{B}_ms_per_tile"] = _stats([t * 1e3 / B for t in ts])
    p = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and "running" not in k else v.clone()) for k, v in sd.items()}
    params = [v for v in p.values() if v.is_floating_point() and v.requires_grad]
    opt = torch.optim.AdamW(params, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))

    def train_fn(B):
        inp, img, lidar = inputs(B)

        def one():
            logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], img, lidar, cfg=cfgv, training=True)
            loss, _, _ = O.pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])
            opt.zero_grad(set_to_none=True)
            loss.backward()
            opt.step()
        return one
    # train step: ONE batch size (B = 4: r02 measured 0.13 tiles/s at B = 1 and 0.20 at B = 8 on the GPU box's host, i.e. 8 - 40 s per
    # iteration), warmed by one B = 1 step, then AT LEAST 5 timed iterations so that `value` is a median of five samples
    train_fn(1)()
    BT = 4
    ts = run(train_fn(BT), 0, 8, 60.0, min_n=5)
    train[f"B{BT}_tiles_per_s"] = {"max": round(BT / min(ts), 4), "median": round(BT / statistics.median(ts), 4), "n": len(ts)}
    best = BT / statistics.median(ts)
    # literal greedy decode (Decoder.predict re-runs the whole padded sequence every step, model_pix2poly.py:187-219): every step costs
    # the same, so a bounded number of steps is timed and scaled to the 385 steps of one tile
    inp, img, lidar = inputs(1)
    with torch.no_grad():
        enc = O.encoder_fusion(img, lidar[0], lidar[1], sd, cfg=cfgv) if kind == "fusion" else (
            O.encoder_vit(img, sd, cfg=cfgv) if kind == "image" else O.encoder_lidar(lidar[0], lidar[1], sd, cfg=cfgv))
        n_dec = 12
        t0 = time.time()
        O.greedy_generate(enc, sd, steps=n_dec)
        dec_s = (time.time() - t0) / n_dec * (O.MAX_LEN - 1)
    return {"value": round(best, 4), "unit": "tiles/s", "cores": torch.get_num_threads(), "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": "oracle (fp32 torch CPU restatement): train step fwd+CE+10*BCE+bwd+AdamW at B = 4 (value = median of >= 5 timed iterations), forward at B = 1 and 8, "
                      f"literal greedy decode ({n_dec} of {O.MAX_LEN - 1} steps timed at B = 1, scaled); wall-clock bounded legs, iteration counts in `n`",
            "forward": fwd, "train": train, "decode_s_per_tile": round(dec_s, 2), "_vs_oracle": vs}


def pmc_step_traffic(precision="bf16"):
    """HBM bytes per train step and per launch of each kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE,
    separate runs of `bench.py --graph 0`; FETCH doubled per the gfx950 note in MI355X_MICROARCH.md; tools/pmc_traffic.py).  bench.py cannot
    collect counters on itself: these are the numbers of the same kernels on the same workload from profiles/."""
    for name in PMC_FILES.get(precision, ()):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as fh:
                return json.load(fh), f"profiles/{name}"
    return None, None


def norm_kernel(name):
    """one spelling for a kernel name from rocprofv3 (unsigned short), tools/kstats.py (bf16) and p3_last_kernel (bf16)"""
    return name.replace("unsigned short", "bf16").replace(" ", "")


def kernel_traffic(table, label):
    if table is None:
        return None
    tot = n = 0.0
    for k, v in table.items():
        if not isinstance(v, dict) or "launches" not in v:
            continue
        parts = [x.strip() for x in k[k.find("<") + 1:k.rfind(">")].split(",")] if "<" in k else []
        hit = False
        if label.startswith("gemm_kernel<bf16,plain"):          # the r01 / r02 label: every plain-A bf16 instantiation
            hit = k.startswith("gemm_kernel<unsigned short,") and len(parts) >= 3 and parts[2] == "0"
        else:                                                    # r03: the label IS the kernel's name (rocprofv3 spelling, bf16 for unsigned short)
            hit = norm_kernel(k) == norm_kernel(label)
        if hit:
            tot += (v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"]
            n += v["launches"]
    return round(tot / n) if n else None


def roofline_leg(args, st, pool, hip, peak_tf):
    """dominant-kernel timing with HIP events on the launch stream: instrumented eager TRAIN steps after the timed region (r03: the step, not
    the forward alone - the rocprofv3 summary under profiles/ is of train steps, and the forward's plain GEMMs now run on three kernels).
    Every p3_gemm / p3_gemm_tn / attention-forward launch is bracketed and labelled with the device kernel the library picked
    (p3_last_kernel), spelled as rocprofv3 prints it."""
    from pixelspointspolygons_amd import ops as _ops
    was_sync, _ops.SYNC_BN[0] = _ops.SYNC_BN[0], False     # rank-0-only leg: no collectives here
    st.step_eager(pool[0])                                  # untimed: allocator and caches warm for the eager form
    hip.KTIMER.enable()
    for i in range(2):
        st.step_eager(pool[(i + 1) % len(pool)])
    torch.cuda.synchronize()
    kt = hip.KTIMER.summary()
    hip.KTIMER.disable()
    _ops.SYNC_BN[0] = was_sync
    if not kt:
        return None
    name, rec = max(kt.items(), key=lambda kv: kv[1]["ms"])
    ranked = sorted(kt.items(), key=lambda kv: -kv[1]["ms"])
    peak_gb = 8000.0
    sec = rec["ms"] * 1e-3
    ach_tf = rec["flop"] / sec / 1e12
    ach_gb = rec["bytes"] / sec / 1e9 if rec["bytes"] else 0.0
    ai = rec["flop"] / rec["bytes"] if rec["bytes"] else float("inf")
    ridge = peak_tf * 1e12 / (peak_gb * 1e9)
    table, tsrc = pmc_step_traffic(args.precision)
    step_traffic = None
    if table:       # sum over all kernels of (FETCH x 2 + WRITE) x launches / profiled train steps (r01 passes: 2 warm-up + 4 timed = 6 steps)
        tot = sum((v["fetch_bytes_corrected"] + v["write_bytes"]) * v["launches"] for v in table.values() if isinstance(v, dict) and "launches" in v)
        step_traffic = round(tot / float(table.get("_steps", 6)))
    def brief(nm, r):
        t = r["ms"] * 1e-3
        return {"kernel": nm, "launches": r["n"], "avg_launch_us": round(r["ms"] * 1e3 / r["n"], 2),
                "mfma_frac": round(r["flop"] / t / 1e12 / peak_tf, 4), "hbm_frac": round(r["bytes"] / t / 1e9 / peak_gb, 4) if r["bytes"] else None}
    common = {"kernel": name, "launches": rec["n"], "avg_launch_us": round(rec["ms"] * 1e3 / rec["n"], 2),
              "measured_over": "2 eager train steps (HIP events around every p3_gemm / p3_gemm_tn / attention-forward launch)",
              "share_of_bracketed_kernel_time": round(rec["ms"] / sum(r["ms"] for r in kt.values()), 3),
              "next_kernels": [brief(nm, r) for nm, r in ranked[1:6]],
              "arithmetic_intensity_flop_per_byte": round(ai, 1), "ridge_flop_per_byte": round(ridge, 1),
              "traffic": kernel_traffic(table, name), "traffic_source": tsrc,
              "algorithmic_bytes_per_launch": round(rec["bytes"] / rec["n"]), "algorithmic_flop_per_launch": round(rec["flop"] / rec["n"]),
              "step_traffic_bytes": step_traffic, "step_algorithmic_bytes": round(STEP_ALGO_BYTES.get(args.workload, 0)) or None}
    # the bound is the one the launch mix sits under (ridge = 2.5 PF / 8 TB/s = 312 FLOP/B for bf16); both fractions are reported
    if ai < ridge:
        return {"bound": "hbm", "achieved": round(ach_gb, 1), "peak": peak_gb, "unit": "GB/s", "frac": round(ach_gb / peak_gb, 4),
                "mfma_achieved_tflops": round(ach_tf, 2), "mfma_frac": round(ach_tf / peak_tf, 4), **common}
    return {"bound": "mfma", "achieved": round(ach_tf, 2), "peak": peak_tf, "unit": "TFLOP/s", "frac": round(ach_tf / peak_tf, 4),
            "hbm_achieved_gbs": round(ach_gb, 1), "hbm_frac": round(ach_gb / peak_gb, 4), **common}


def encoder_fwd_leg(args, model, st, pool, hip, kind, peak_tf):
    """The quantity north_star's target is defined on: the fused ViT + LiDAR ENCODER forward alone (both stems -> fusion conv + BN + ReLU ->
    12 blocks -> final norm -> channel pool; early_fusion_vit.py:96-127) at the bench batch, train-mode BatchNorm, no autograd, one hipGraph.
    TFLOP/s on SURVEY §8d's count for this part (49.1 GFLOP / tile at 3 k points), + its top kernels from an instrumented eager pass."""
    enc = model.encoder

    def run():
        s_ = st.static
        with torch.no_grad():
            if kind == "fusion":
                return enc(s_["image"], st._lidar())
            return enc(s_["image"]) if kind == "image" else enc(st._lidar())
    st.load(pool[0])
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    graph = None
    if st.use_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run()
    n = max(5, min(args.steps, 20))
    reps = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            graph.replay() if graph is not None else run()
        torch.cuda.synchronize()
        reps.append((time.perf_counter() - t0) / n * 1e3)
    ms = statistics.median(reps)
    gf = GFLOP_ENC[args.workload]
    tf = gf * args.batch / (ms * 1e-3) / 1e3
    out = {"ms_per_batch": round(ms, 3), "ms_per_tile": round(ms / args.batch, 4), "ms_per_batch_all": [round(r, 3) for r in reps], "tiles_per_s": round(args.batch / ms * 1e3, 1),
           "gflop_per_tile": gf, "tflops": round(tf, 1), "mfma_frac": round(tf / peak_tf, 4), "mfma_peak_tflops": peak_tf, "hip_graph": graph is not None,
           "what": "encoder only: patch-embed + pillar stem -> fusion conv + BN + ReLU -> 12 ViT blocks -> norm -> pool; train-mode BatchNorm, no autograd"}
    if not args.no_kernel_timing:
        hip.KTIMER.enable()
        run()
        torch.cuda.synchronize()
        kt = hip.KTIMER.summary()
        hip.KTIMER.disable()
        tot = sum(r["ms"] for r in kt.values()) or 1.0
        out["top_kernels"] = [{"kernel": nm, "launches": r["n"], "avg_launch_us": round(r["ms"] * 1e3 / r["n"], 2), "ms": round(r["ms"], 3),
                               "mfma_frac": round(r["flop"] / (r["ms"] * 1e-3) / 1e12 / peak_tf, 4) if r["ms"] > 0 else None}
                              for nm, r in sorted(kt.items(), key=lambda kv: -kv[1]["ms"])[:3]]
        out["bracketed_ms_eager"] = round(tot, 3)
    return out


def parity_probe(model, pool, kind, n=2):
    """Device side of `<dtype>_vs_oracle`: eval-mode forward of the FIRST n tiles of the bench batch through the bench's own model (its weights
    after the timed steps), with everything the oracle needs to repeat it on the host (inputs, state_dict) copied to host memory."""
    b = pool[0]
    was = model.training
    model.eval()
    with torch.no_grad():
        img = b["image"][:n] if kind != "lidar" else None
        lidar = None
        if kind != "image":
            off = b["lidar_offsets"][: n + 1]
            lidar = (b["lidar_values"][: int(off[-1])].contiguous(), off.contiguous())
        y = b["y"][:n, :-1]
        logits, perm = model(img, lidar, y)
    torch.cuda.synchronize()
    model.train(was)
    cpu = lambda t: None if t is None else t.detach().float().cpu() if t.is_floating_point() else t.detach().cpu()
    return {"logits": cpu(logits), "perm": cpu(perm), "y": cpu(y), "image": cpu(img), "lidar": None if lidar is None else (cpu(lidar[0]), cpu(lidar[1])),
            "sd": {k: (v.detach().float().cpu() if v.is_floating_point() else v.detach().cpu()) for k, v in model.state_dict().items()}}


def dense_lidar_leg(args, model, S, dev, rank):
    """SURVEY §8d's dense variant: the LiDAR stem alone (pillarize + PillarFeatureNet + scatter, pointpillars_o3d.py:85-107) at the density of
    real tiles (40 000 points per tile, predictor.py:120-133: ~51 points per pillar, the cap of 64 binds in a share of them) beside the bench's
    3 k-point clouds; HIP events on the launch stream, train-mode BatchNorm, no autograd.  HBM-bound sub-kernel: GB/s on its algorithmic bytes
    (12 B per point in + the [B, 784, C] canvas out)."""
    stem = getattr(model.encoder, "lidar_embed", None)
    if stem is None:
        return None
    out = {"tiles": args.batch, "max_points_per_pillar": stem.max_points,
           "what": "PointPillarsEncoder.forward alone: counting sort + PFN layer 0 / layer-1 GEMM / max + scatter; algorithmic bytes = points in + canvas out"}
    for label, npts in (("bench_3k", args.points), ("dense_40k", 40000)):
        inp = S.make_inputs(args.batch, seed=777 + rank, n_points=npts, jitter=npts // 10)
        vals, offs = inp["lidar_values"].to(dev), inp["lidar_offsets"].to(dev)
        with torch.no_grad():
            for _ in range(3):
                c = stem((vals, offs), return_flattened=True)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 10
            e0.record()
            for _ in range(n):
                c = stem((vals, offs), return_flattened=True)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        nbytes = vals.numel() * 4 + c.numel() * c.element_size()
        out[label] = {"points_per_tile": npts, "ms_per_batch": round(ms, 3), "mpoints_per_s": round(vals.shape[0] / ms / 1e3, 1),
                      "algorithmic_bytes": int(nbytes), "hbm_gbs": round(nbytes / ms / 1e6, 1), "hbm_frac": round(nbytes / ms / 1e6 / 8000.0, 4)}
        del vals, offs, c
    return out


def ffl_leg(args, dev, local, S, rank, world):
    """BASELINE configs[4] driver-timed: FFL early_fusion_vit_cnn at ITS real size (ViT depth 12, 224 x 224 heads, bs 64), a short run of the same
    step `--workload ffl_fusion` times (forward + FFL criterion + backward + AdamW under one hipGraph)."""
    import argparse as _ap
    a2 = _ap.Namespace(**vars(args))
    a2.workload = "ffl_fusion"
    _, model, opt, reducer, pool, st = build(a2, dev, local, args.precision, S, rank, world, False)
    n = 5
    dt, loss = timed_steps(st, pool, n, 3, 1, dev)
    out = {"value": round(args.batch * n / dt, 2), "unit": "tiles/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n, "warmup": 3, "dtype": DTYPE_NAME[args.precision],
           "workload": f"ffl_early_fusion_vit_cnn_bs{args.batch}x1", "step": "fwd+FFL criterion+bwd+AdamW", "hip_graph": st.graph is not None,
           "gflop_fwd_per_tile": GFLOP_FWD["ffl_fusion"], "step_mfma_frac": round(3 * GFLOP_FWD["ffl_fusion"] * args.batch / (dt / n) / 1e3 / PEAK_TF[args.precision], 4), "mfma_peak_tflops": round(PEAK_TF[args.precision], 1),
           "final_loss": round(loss, 4)}
    opt.close()
    del st, reducer, opt, model, pool
    torch.cuda.empty_cache()
    return out


def image_b16_leg(args, dev, local, S, rank, world, with_oracle):
    """BASELINE configs[1] driver-timed: image-only Pix2Poly with the ViT-B/16 shape (197 tokens x 768, depth 12), bs 64 - the train step in `bf16` (the
    precision configs[1] names) and in `fp32x3` (the one that meets north_star's tolerance), each with its own measured error against the oracle."""
    import argparse as _ap
    a2 = _ap.Namespace(**vars(args))
    a2.workload = "image_b16"
    out = {"workload": f"pix2poly_image_b16_bs{args.batch}x1", "step": "fwd+CE+10*BCE+bwd+AdamW", "gflop_fwd_per_tile_executed": GFLOP_FWD["image_b16"]}
    probes = {}
    for prec in ("bf16", "fp32x3"):
        _, model, opt, reducer, pool, st = build(a2, dev, local, prec, S, rank, world, False)
        n = max(3, min(args.steps, 10))
        dt, loss = timed_steps(st, pool, n, 3, 1, dev)
        out[prec] = {"value": round(args.batch * n / dt, 2), "unit": "tiles/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n, "warmup": 3, "dtype": DTYPE_NAME[prec],
                     "hip_graph": st.graph is not None, "mfma_peak_tflops": round(PEAK_TF[prec], 1),
                     "step_mfma_frac": round(3 * GFLOP_FWD["image_b16"] * args.batch / (dt / n) / 1e3 / PEAK_TF[prec], 4), "final_loss": round(loss, 4)}
        if with_oracle:
            probes[f"image_b16_{prec}_vs_oracle"] = parity_probe(model, pool, "image")
        opt.close()
        del st, reducer, opt, model, pool
        torch.cuda.empty_cache()
    return out, probes


def build(args, dev, local, precision, S, rank, world, sync_bn):
    """model + optimizer + reducer + synthetic pool + stepper for one precision"""
    from pixelspointspolygons_amd.training import FlatAdamW, GradBucketReducer
    from pixelspointspolygons_amd.vision_transformer import compute_dtype
    kind = KIND[args.workload]
    cfg = make_cfg(args, dev, precision=precision)
    torch.manual_seed(42)                    # reference seed (train/trainer.py:214); random-init weights (no checkpoints offline)
    criterion = None
    if args.workload == "ffl_fusion":
        from pixelspointspolygons_amd.ffl import FFLModel
        from pixelspointspolygons_amd.ffl_losses import build_combined_loss
        model = FFLModel(cfg, local)
        criterion = build_combined_loss(cfg)
    else:
        from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
        model = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, local)
        if args.no_dropout:
            model.decoder.set_dropout(0.0)
        # decoder dropout stays at the reference's training defaults (0.1 in nn.TransformerDecoderLayer incl. the attention
        # probabilities, 0.05 on both positional sums, model_pix2poly.py:136-143): fused into the GEMM epilogues / attention kernels
    model.train()
    opt = FlatAdamW(model, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), compute_dtype=compute_dtype(cfg), direct_grad=True)
    opt.set_linear_schedule(200 * 1000)
    reducer = GradBucketReducer(opt)         # N > 1: bucket all-reduces launched from backward as buckets complete (side stream)
    pool = [synth_batch(S, args, rank, s, dev, kind, ffl=criterion is not None) for s in range(args.pool)]
    st = Stepper(model, opt, reducer, pool, kind, bool(args.graph), criterion=criterion)
    return cfg, model, opt, reducer, pool, st


def self_launch(args):
    """`--gpus N` with N > 1 and no launcher around us: become the launcher.  Nothing in this process has initialised the GPU yet (importing
    torch does not), the ranks are CHILD processes (never an exec of a process that may hold the device), their stdout is relayed line by
    line - rank 0's JSON line stays the last one - and their exit code is ours."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=os.getcwd())
    for ln in proc.stdout:
        sys.stdout.write(ln)
        sys.stdout.flush()
    return proc.wait()


def dry_run(args, world, rank, backend):
    """P3_BENCH_DRYRUN=1 (test hook, no GPU needed): the launch / rendezvous / bracket / one-JSON-line control flow with the step replaced
    by a sleep - what the CPU test of `bench.py --gpus 2` runs.  The line says so (`dry_run`: true, value 0)."""
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001)
    if world > 1:
        dist.barrier()
        t = torch.tensor([time.perf_counter() - t0])
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        got = dist.get_world_size()
        dist.destroy_process_group()
    else:
        got = 1
    if rank == 0:
        print(json.dumps({"metric": "training tiles/sec (224px img + 3k-pt lidar)", "value": 0.0, "unit": "tiles/s", "n_gpus": got, "steps": args.steps,
                          "warmup": args.warmup, "dry_run": True, "config": {"workload": f"pix2poly_{args.workload}_bs{args.batch}x{got}",
                                                                            "collectives": {"world": got}, "parallelism": f"dp{got}"}}), flush=True)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and os.environ.get("P3_FORCE_COLLECTIVES") != "1":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks (they must agree)")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks for the 1-GPU development boxes (the N > 1 control flow is exercised with two ranks sharing device 0 over gloo):
    # P3_BENCH_BACKEND=gloo P3_BENCH_ONE_DEVICE=1.  The driver's multi-GPU runs use neither: one rank per GPU over RCCL.
    if os.environ.get("P3_BENCH_ONE_DEVICE") == "1":
        local = 0
    backend = os.environ.get("P3_BENCH_BACKEND", "nccl")
    if os.environ.get("P3_BENCH_DRYRUN") == "1":
        return dry_run(args, world, rank, backend)
    torch.cuda.set_device(local)
    dev = f"cuda:{local}"
    forced = world == 1 and os.environ.get("P3_FORCE_COLLECTIVES") == "1"   # 1-rank RCCL smoke run (ops.SINGLE_RANK_COLLECTIVES)
    if forced:
        for k, v in (("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29617"), ("RANK", "0"), ("WORLD_SIZE", "1")):
            os.environ.setdefault(k, v)
    if world > 1 or forced:
        if backend == "nccl":
            dist.init_process_group("nccl", init_method="env://", device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, init_method="env://")
        assert dist.get_world_size() == (args.gpus if not forced else 1), (dist.get_world_size(), args.gpus)
        warm = torch.zeros(1, device=dev)
        dist.all_reduce(warm)                             # communicator set-up (and RCCL's banner) happen here, before anything is timed
        torch.cuda.synchronize()
        _flush_c_stdio()
    from pixelspointspolygons_amd import synthetic as S        # synthetic-input generator (oracle/ is imported by the cpu_baseline leg only)
    from pixelspointspolygons_amd import hip, ops

    kind = KIND[args.workload]
    # N > 1 = the reference's DDP setup (model_pix2poly.py:324-328, model_ffl.py:161-163): SyncBatchNorm in every BatchNorm site (statistics
    # packed into one buffer per module group) + gradient buckets all-reduced on a side stream while backward continues
    sync_bn = (world > 1 or forced) and bool(args.sync_bn)
    ops.SYNC_BN[0] = sync_bn
    cfg, model, opt, reducer, pool, st = build(args, dev, local, args.precision, S, rank, world, sync_bn)
    dt, loss_val = timed_steps(st, pool, args.steps, max(args.warmup, 3 if st.use_graph else 0), world, dev)   # 2 eager steps + the capture stay untimed

    # forward-only latency (train-mode forward of the same model state, no grad; hipGraph)
    fwd_ms = fwd_eval_ms = float("nan")
    if not args.no_fwd and not reducer.active:       # N = 1 legs (single-GPU latency figures; no extra collectives in a scaling run)
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

    single = world == 1 and not forced
    rank_spread = timed_steps.rank_spread
    peak_tf = PEAK_TF[args.precision]
    pix = args.workload != "ffl_fusion"
    enc_fwd = dense = None
    probes = {}
    if single and rank == 0 and pix and not args.no_fwd:
        enc_fwd = encoder_fwd_leg(args, model, st, pool, hip, kind, peak_tf)
        dense = dense_lidar_leg(args, model, S, dev, rank)
    if single and rank == 0 and pix and not args.no_cpu_baseline:
        probes[f"{args.precision}_vs_oracle"] = parity_probe(model, pool, kind)
    opt_buckets, early_launches, sync_calls = list(opt.buckets), reducer.early_launches, ops.SYNC_CALLS[0]
    graph_used = st.graph is not None
    feed = None
    if single and not args.no_host_feed and args.workload != "ffl_fusion":
        feed = host_feed_leg(S, args, st, dev, kind, rank)

    roofline = None
    if rank == 0 and not args.no_kernel_timing:
        roofline = roofline_leg(args, st, pool, hip, peak_tf)

    # the same step in the other two precisions: bf16 storage (faster; its measured error says why it is not the headline) and the exact fp32 MFMA path
    other_legs = {}
    main_alive = True                        # the headline model / optimizer / stepper still exist
    if single and not args.no_fp32_leg:
        main_alive = False
        del st, reducer
        opt.close()
        del opt, model
        torch.cuda.empty_cache()
        whats = {"bf16": "precision='bf16': bf16 storage of activations and of a shadow copy of the weights, fp32 accumulate / LayerNorm / softmax / loss / master weights; its own measured "
                         "error against the oracle is `bf16_vs_oracle` (1e-2, argmax agreement < 1): outside north_star's tolerance, hence a sub-object and not `value`",
                 "fp32": "precision='fp32': v_mfma_f32_32x32x2_f32 everywhere (bit-exact fmaf chains)",
                 "fp32x3": "precision='fp32x3': fp32 storage, every product as bf16 x 3 (2^-17 per product)"}
        for prec in [p_ for p_ in ("bf16", "fp32", "fp32x3") if p_ != args.precision]:
            _, model_o, opt_o, _, pool_o, st_o = build(args, dev, local, prec, S, rank, world, False)
            n_o = max(3, min(args.steps, 20 if prec == "bf16" else 6))
            dt_o, loss_o = timed_steps(st_o, pool_o, n_o, 3, 1, dev)
            if rank == 0 and pix and not args.no_cpu_baseline:
                probes[f"{prec}_vs_oracle"] = parity_probe(model_o, pool_o, kind)
            other_legs[prec] = {"value": round(args.batch * n_o / dt_o, 2), "unit": "tiles/s", "ms_per_step": round(dt_o / n_o * 1e3, 3), "steps": n_o, "warmup": 3,
                                "dtype": DTYPE_NAME[prec], "mfma_peak_tflops": round(PEAK_TF[prec], 1),
                                "step_mfma_frac": round(3 * GFLOP_FWD[args.workload] * args.batch / (dt_o / n_o) / 1e3 / PEAK_TF[prec], 4),
                                "final_loss": round(loss_o, 4), "what": whats[prec]}
            opt_o.close()
            del st_o, opt_o, model_o, pool_o
            torch.cuda.empty_cache()

    ffl = None
    if single and rank == 0 and pix and not args.no_ffl and args.workload == "fusion_s8":
        if main_alive:
            main_alive = False
            del st, reducer
            opt.close()
            del opt, model
            torch.cuda.empty_cache()
        ffl = ffl_leg(args, dev, local, S, rank, world)

    b16 = None
    if single and rank == 0 and pix and not args.no_b16 and args.workload == "fusion_s8":
        if main_alive:
            main_alive = False
            del st, reducer
            opt.close()
            del opt, model
            torch.cuda.empty_cache()
        b16, b16_probes = image_b16_leg(args, dev, local, S, rank, world, not args.no_cpu_baseline)
        probes.update(b16_probes)

    predict = None
    if single and rank == 0 and not args.no_predict and args.workload != "ffl_fusion":
        try:
            from pixelspointspolygons_amd.predict_demo import predict_leg
            predict = predict_leg(dev)
        except ImportError:
            predict = None

    if rank == 0:
        tiles = args.batch * world * args.steps
        gf = GFLOP_FWD[args.workload]
        step_s = dt / args.steps
        line = {
            "metric": "training tiles/sec (224px img + 3k-pt lidar)", "value": round(tiles / dt, 2), "unit": "tiles/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(step_s * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_NAME[args.precision], "precision": args.precision, "data": "synthetic",
            "config": {"workload": (f"ffl_early_fusion_vit_cnn_bs{args.batch}x{world}" if args.workload == "ffl_fusion" else f"pix2poly_{args.workload}_bs{args.batch}x{world}"), "tiles_per_gpu": args.batch,
                       "points_per_tile": args.points, "hip_graph": graph_used, "sync_bn": sync_bn,
                       **({"collectives": {"backend": dist.get_backend(), "world": world, "forced_single_rank": forced, "grad_buckets": len(opt_buckets),
                                           "early_bucket_launches": early_launches, "syncbn_collectives": sync_calls}} if (world > 1 or forced) else {}),
                       "decoder_dropout": "off (A/B run)" if args.no_dropout else "reference defaults (0.1 / 0.05)",
                       "step": "fwd+FFL criterion+bwd+AdamW" if args.workload == "ffl_fusion" else "fwd+CE+10*BCE+bwd+AdamW", "parallelism": f"dp{world}"},
            "fwd_ms_per_tile": round(fwd_ms / args.batch, 4) if fwd_ms == fwd_ms else None, "fwd_ms_per_batch": round(fwd_ms, 3) if fwd_ms == fwd_ms else None,
            "fwd_eval_ms_per_tile": round(fwd_eval_ms / args.batch, 4) if fwd_eval_ms == fwd_eval_ms else None,
            "gflop_fwd_per_tile_executed": gf, "gflop_fwd_per_tile_dense_reference": GFLOP_FWD_DENSE[args.workload],
            "fwd_mfma_frac": round(gf * args.batch / (fwd_ms * 1e-3) / 1e3 / peak_tf, 4) if fwd_ms == fwd_ms else None,
            "step_mfma_frac": round(3 * gf * args.batch / step_s / 1e3 / peak_tf, 4),
            "mfma_peak_tflops": round(peak_tf, 1),
            "mfma_peak_what": {"bf16": "bf16 dense MFMA", "fp32x3": "bf16 dense MFMA / 3: every algorithmic product issues three bf16 MFMAs (a_lo b_hi + a_hi b_lo + a_hi b_hi)",
                               "fp32": "fp32-input MFMA (v_mfma_f32_32x32x2_f32)"}[args.precision],
            "final_loss": round(loss_val, 4),
            "roofline": roofline,
        }
        if enc_fwd is not None:
            line["encoder_fwd"] = enc_fwd
        if dense is not None:
            line["dense_lidar"] = dense
        if rank_spread is not None:
            line["rank_ms_per_step"] = {"min": round(rank_spread[0] / args.steps * 1e3, 3), "max": round(rank_spread[1] / args.steps * 1e3, 3)}
        for prec, leg in other_legs.items():
            line[{"bf16": "bf16_mode", "fp32": "fp32_exact_mode", "fp32x3": "fp32x3_mode"}[prec]] = leg
        if ffl is not None:
            line["ffl"] = ffl
        if b16 is not None:
            line["image_b16"] = b16
        if predict is not None:
            line["predict"] = predict
        if feed is not None:
            line["pcie_inclusive"] = feed
        if not args.no_cpu_baseline and single and args.workload != "ffl_fusion":   # N = 1 only (the ranks would share the host cores)
            cb = cpu_baseline(args, kind, probes)
            for k_, v_ in cb.pop("_vs_oracle").items():
                line[k_] = v_
            line["cpu_baseline"] = cb
    if world > 1 or forced:
        dist.barrier()                                    # rank 0 runs the instrumented leg alone: leave together
        dist.destroy_process_group()
    _flush_c_stdio()
    if rank == 0:
        print(json.dumps(line), flush=True)               # the LAST line of stdout


def _flush_c_stdio():
    """RCCL writes a version banner to the C stdout buffer at init; unflushed it would land AFTER the JSON line at exit."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except OSError:
        pass


if __name__ == "__main__":
    main()
