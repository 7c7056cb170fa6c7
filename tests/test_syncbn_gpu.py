"""SyncBatchNorm + data-parallel gradient averaging (model_pix2poly.py:326-328) on the HIP path: two ranks, each with
half of a batch, must reproduce the oracle run on the WHOLE batch (joint BatchNorm statistics in all 9 BatchNorm sites, forward and
backward).  With two visible GPUs the ranks use one device each over RCCL ("nccl"); on the 1-GPU development boxes they share device 0
and talk over gloo (the process group stages device tensors through the host by itself) - same code path above the backend."""
import os
import socket
import sys
import tempfile

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _init(rank, world, port):
    import datetime
    import faulthandler
    import torch.distributed as dist
    # a rank that dies (or waits for one that died) must not hold the suite for the process group's default 30 minutes: collectives time out after 4 minutes and a
    # rank still alive after 6 prints every thread's stack and exits (r04: one full-suite run sat in exactly this state until the box's time limit)
    faulthandler.dump_traceback_later(360, exit=True)
    tmo = datetime.timedelta(seconds=240)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if torch.cuda.device_count() >= world:
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"), timeout=tmo)
        return f"cuda:{rank}"
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=tmo)
    torch.cuda.set_device(0)
    return "cuda:0"


def _spawn(fn, world, outdir, *extra):
    """two ranks on this box; a rendezvous / transport failure (one rank never connects: the collectives' 4-minute timeout or the 6-minute watchdog of _init
    fires) is retried ONCE on a fresh port - a second failure is the test's failure"""
    import torch.multiprocessing as mp
    for attempt in (0, 1):
        try:
            mp.spawn(fn, args=(world, _free_port(), outdir, *extra), nprocs=world, join=True)
            return
        except Exception as e:                      # noqa: BLE001 - reported below
            if attempt == 1:
                raise
            print(f"[test_syncbn_gpu] first two-rank launch failed ({type(e).__name__}: {str(e)[:300]}); retrying once", flush=True)


def _worker(rank, world, port, outdir, precision):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import pix2poly_loss
    dev = _init(rank, world, port)
    try:
        sd = O.make_state_dict("fusion", seed=42)
        inp = O.make_inputs(2 * world, seed=99)
        cfg = make_config("early_fusion_vit", precision=precision, device=dev)
        m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
        m.load_state_dict(sd, strict=True)
        m.train()
        m.decoder.set_dropout(0.0)
        ops.SYNC_BN[0] = True
        lo, hi = 2 * rank, 2 * rank + 2
        off = inp["lidar_offsets"]
        vals = inp["lidar_values"][off[lo]:off[hi]].to(dev)
        offs = (off[lo:hi + 1] - off[lo]).to(dev)
        y = inp["y"][lo:hi].to(dev)
        logits, perm = m(inp["image"][lo:hi].to(dev), (vals, offs), y[:, :-1])
        loss, _, _ = pix2poly_loss(logits, perm, y[:, 1:], inp["y_perm"][lo:hi].to(dev))
        loss.backward()
        grads = {}
        for k, p in m.named_parameters():          # what the DDP reducer does: average over ranks
            g = p.grad.float().cpu()
            dist.all_reduce(g)
            grads[k] = g / world
        torch.save(dict(logits=logits.detach().float().cpu(), perm=perm.detach().float().cpu(), loss=float(loss), grads=grads,
                        rmean=m.encoder.fusion_layer[1].running_mean.cpu()), os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
def test_two_ranks_with_sync_batchnorm_equal_the_oracle_on_the_whole_batch(precision):
    import torch.multiprocessing as mp
    from oracle import p3_oracle as O
    from tests.helpers import l2_err, rel_err
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        _spawn(_worker, world, outdir, precision)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    sd = O.make_state_dict("fusion", seed=42)
    inp = O.make_inputs(2 * world, seed=99)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], inp["image"].double(), (inp["lidar_values"], inp["lidar_offsets"]), training=True)
    losses = [O.pix2poly_loss(logits[2 * r:2 * r + 2], perm[2 * r:2 * r + 2], inp["y"][2 * r:2 * r + 2, 1:], inp["y_perm"][2 * r:2 * r + 2].double())[0]
              for r in range(world)]
    (sum(losses) / world).backward()
    for r in range(world):
        assert rel_err(res[r]["logits"], logits[2 * r:2 * r + 2].detach()) < 1e-3
        assert rel_err(res[r]["perm"], perm[2 * r:2 * r + 2].detach()) < 1e-3
        assert abs(res[r]["loss"] - float(losses[r])) < 2e-3 * abs(float(losses[r]))
    assert torch.equal(res[0]["rmean"], res[1]["rmean"])        # identical running statistics on both ranks
    assert rel_err(res[0]["rmean"], p["encoder.fusion_layer.1.running_mean"]) < 1e-4
    gnorm = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad)
    bad = {}
    for k, g in res[0]["grads"].items():
        e = l2_err(g, p[k].grad, floor=1e-3 * gnorm)
        if not e < 1.5e-2:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


def _worker_factory_wrap(rank, world, port, outdir, precision):
    """what an UNCHANGED train.py does on N > 1 GPUs: cfg.host.multi_gpu=True -> the factory returns DistributedDataParallel(convert_sync_batchnorm(model))
    (model_pix2poly.py:324-328), the trainer puts stock nn.CrossEntropyLoss / nn.BCELoss and torch.optim.AdamW on top (trainer_pix2poly.py:38-93,316-329)"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import torch.nn as nn
    from torch.nn.parallel import DistributedDataParallel as DDP
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    dev = _init(rank, world, port)
    try:
        sd = O.make_state_dict("fusion", seed=42)
        inp = O.make_inputs(2 * world, seed=99)
        cfg = make_config("early_fusion_vit", precision=precision, device=dev, multi_gpu=True)
        tk = Tokenizer(cfg)
        m = Pix2PolyModel(cfg, tk.vocab_size, local_rank=torch.cuda.current_device())
        assert isinstance(m, DDP) and not any(type(x) is nn.BatchNorm2d or type(x) is nn.BatchNorm1d for x in m.modules())
        assert sum(isinstance(x, nn.SyncBatchNorm) for x in m.modules()) == 9
        m.module.load_state_dict(sd, strict=True)
        m.train()
        m.module.decoder.set_dropout(0.0)
        ce, bce = nn.CrossEntropyLoss(ignore_index=cfg.experiment.model.tokenizer.pad_idx), nn.BCELoss()
        opt = torch.optim.AdamW(m.parameters(), lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))
        lo, hi = 2 * rank, 2 * rank + 2
        off = inp["lidar_offsets"]
        vals = inp["lidar_values"][off[lo]:off[hi]].to(dev)
        offs = (off[lo:hi + 1] - off[lo]).to(dev)
        y = inp["y"][lo:hi].to(dev)
        preds, perm = m(inp["image"][lo:hi].to(dev), torch.nested.nested_tensor_from_jagged(vals, offs), y[:, :-1])
        y_expected = y[:, 1:]
        loss = 1.0 * ce(preds.reshape(-1, preds.shape[-1]), y_expected.reshape(-1)) + 10.0 * bce(perm, inp["y_perm"][lo:hi].to(dev))
        opt.zero_grad(set_to_none=True)
        loss.backward()                                  # DDP's reducer averages the gradients over the ranks
        grads = {k: p.grad.float().cpu() for k, p in m.module.named_parameters()}
        opt.step()
        after = {k: dict(m.module.named_parameters())[k].detach().float().cpu() for k in ("decoder.output.weight", "encoder.vit.blocks.0.attn.qkv.weight", "bin_score")}
        torch.save(dict(logits=preds.detach().float().cpu(), perm=perm.detach().float().cpu(), loss=float(loss), grads=grads, after=after,
                        rmean=m.module.encoder.fusion_layer[1].running_mean.cpu(), rvar=m.module.scorenet2.bn3.running_var.cpu()), os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("precision", ["fp32x3"])
def test_reference_factory_wrap_sync_batchnorm_plus_torch_ddp_equals_the_oracle_on_the_whole_batch(precision):
    """SURVEY a-13 as the reference wires it (model_pix2poly.py:324-328, trainer_pix2poly.py:316-329): Pix2PolyModel(cfg with host.multi_gpu=True) ->
    convert_sync_batchnorm + torch DistributedDataParallel, stock CrossEntropyLoss / BCELoss / torch.optim.AdamW on top, two ranks with half a batch each:
    forward, joint BatchNorm statistics, the reducer's averaged gradients and one AdamW step equal the float64 oracle on the WHOLE batch."""
    from oracle import p3_oracle as O
    from tests.helpers import l2_err, rel_err
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        _spawn(_worker_factory_wrap, world, outdir, precision)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    sd = O.make_state_dict("fusion", seed=42)
    inp = O.make_inputs(2 * world, seed=99)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], inp["image"].double(), (inp["lidar_values"], inp["lidar_offsets"]), training=True)
    losses = [O.pix2poly_loss(logits[2 * r:2 * r + 2], perm[2 * r:2 * r + 2], inp["y"][2 * r:2 * r + 2, 1:], inp["y_perm"][2 * r:2 * r + 2].double())[0]
              for r in range(world)]
    (sum(losses) / world).backward()
    for r in range(world):
        assert rel_err(res[r]["logits"], logits[2 * r:2 * r + 2].detach()) < 1e-3
        assert rel_err(res[r]["perm"], perm[2 * r:2 * r + 2].detach()) < 1e-3
        assert abs(res[r]["loss"] - float(losses[r])) < 2e-3 * abs(float(losses[r]))
    assert torch.equal(res[0]["rmean"], res[1]["rmean"]) and torch.equal(res[0]["rvar"], res[1]["rvar"])
    assert rel_err(res[0]["rmean"], p["encoder.fusion_layer.1.running_mean"]) < 1e-4
    assert rel_err(res[0]["rvar"], p["scorenet2.bn3.running_var"]) < 1e-3
    gnorm = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad)
    bad = {}
    for k, g in res[0]["grads"].items():
        assert torch.equal(g, res[1]["grads"][k]), k                  # every rank holds the reducer's average
        e = l2_err(g, p[k].grad, floor=1e-3 * gnorm)
        if not e < 1.5e-2:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    # one stock AdamW step from those gradients: the parameters of both ranks move together and by lr * sign-like first-step updates
    ref = {k: v.detach().clone().requires_grad_(True) for k, v in sd.items() if k in res[0]["after"]}
    ropt = torch.optim.AdamW(list(ref.values()), lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95))
    for k, v in ref.items():
        v.grad = p[k].grad.float()
    ropt.step()
    for k, v in res[0]["after"].items():
        assert torch.equal(v, res[1]["after"][k]), k
        step = (v - sd[k]).abs().max()
        assert float(step) > 1e-4 and float(step) < 4e-4, (k, float(step))          # first AdamW step: |update| ~ lr
        agree = ((v - sd[k]).sign() == (ref[k].detach() - sd[k]).sign()).float().mean()
        assert float(agree) > 0.98, (k, float(agree))                    # same update direction as AdamW on the oracle's gradient (sign flips only where g ~ 0)


def _worker_reducer(rank, world, port, outdir):
    """the bench's N > 1 step shape: FlatAdamW(direct_grad) + GradBucketReducer, SyncBatchNorm on, two identical steps"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, GradBucketReducer, pix2poly_loss
    dev = _init(rank, world, port)
    try:
        sd = O.make_state_dict("fusion", seed=42)
        inp = O.make_inputs(2 * world, seed=99)
        cfg = make_config("early_fusion_vit", precision="fp32", device=dev)
        m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
        m.load_state_dict(sd, strict=True)
        m.train()
        m.decoder.set_dropout(0.0)
        ops.SYNC_BN[0] = True
        opt = FlatAdamW(m, compute_dtype=torch.float32, bucket_mb=8, direct_grad=True)
        red = GradBucketReducer(opt)
        lo, hi = 2 * rank, 2 * rank + 2
        off = inp["lidar_offsets"]
        vals, offs = inp["lidar_values"][off[lo]:off[hi]].to(dev), (off[lo:hi + 1] - off[lo]).to(dev)
        y = inp["y"][lo:hi].to(dev)
        img, yp = inp["image"][lo:hi].to(dev), inp["y_perm"][lo:hi].to(dev)
        info = []
        for step in range(2):
            c0 = ops.SYNC_CALLS[0]
            logits, perm = m(img, (vals, offs), y[:, :-1])
            fwd_calls = ops.SYNC_CALLS[0] - c0
            loss, _, _ = pix2poly_loss(logits, perm, y[:, 1:], yp)
            opt.zero_grad()
            loss.backward()
            early = red.early_launches
            scale = red.finish()
            info.append(dict(fwd_calls=fwd_calls, bwd_calls=ops.SYNC_CALLS[0] - c0 - fwd_calls, early=early))
        grads = {k: (p.grad.float() * scale).cpu() for k, p in m.named_parameters()}
        torch.save(dict(grads=grads, info=info, nb=len(opt.buckets)), os.path.join(outdir, f"rank{rank}.pt"))
        red.close()
        opt.close()
    finally:
        dist.destroy_process_group()


def test_bucketed_reducer_overlaps_backward_under_sync_batchnorm():
    """FlatAdamW(direct_grad) + GradBucketReducer + SyncBatchNorm on two ranks: averaged arena gradients == float64 autograd of the oracle
    on the whole batch; from the second step on the buckets are all-reduced from INSIDE backward (launched when the last kernel writing
    a bucket has been enqueued), although the weight gradients bypass autograd's AccumulateGrad; one collective per BatchNorm site."""
    import torch.multiprocessing as mp
    from oracle import p3_oracle as O
    from tests.helpers import l2_err
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        _spawn(_worker_reducer, world, outdir)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    info = res[0]["info"]
    assert res[0]["nb"] >= 3
    assert info[0]["early"] == 0 and info[1]["early"] >= res[0]["nb"] - 1, info          # step 0 records, step 1 overlaps
    # pillar stem 2 (+ pillar count packed with the first) + fusion conv 1 + 3 ScoreNet depths (scorenet1 / scorenet2 advance in lockstep and share
    # one message per depth, r04: 2 x 3 -> 3), forward; the same sites backward: 6 + 6 collectives per step instead of 9 + 9
    assert info[1]["fwd_calls"] == 6 and info[1]["bwd_calls"] <= 6, info
    sd = O.make_state_dict("fusion", seed=42)
    inp = O.make_inputs(2 * world, seed=99)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    logits, perm = O.pix2poly_forward(p, inp["y"][:, :-1], inp["image"].double(), (inp["lidar_values"], inp["lidar_offsets"]), training=True)
    losses = [O.pix2poly_loss(logits[2 * r:2 * r + 2], perm[2 * r:2 * r + 2], inp["y"][2 * r:2 * r + 2, 1:], inp["y_perm"][2 * r:2 * r + 2].double())[0]
              for r in range(world)]
    (sum(losses) / world).backward()
    gnorm = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad)
    for r in range(world):
        bad = {}
        for k, g in res[r]["grads"].items():
            e = l2_err(g, p[k].grad, floor=1e-3 * gnorm)
            if not e < 1.5e-2:
                bad[k] = e
        assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]
    assert all(torch.equal(res[0]["grads"][k], res[1]["grads"][k]) for k in res[0]["grads"])     # every rank holds the same average


SMALL = dict(dim=384, depth=2, heads=6, mlp=1536, patch=8, img=224, eps=1e-6)


def _worker_ffl(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl import FFLModel
    dev = _init(rank, world, port)
    try:
        cfg = make_config("early_fusion_vit_cnn", model="ffl", precision="fp32", vit_depth=2, device=dev)
        m = FFLModel(cfg, 0)
        m.load_state_dict({k: v.clone() for k, v in O.make_ffl_state_dict("fusion", SMALL, seed=11).items()})
        m.train()
        ops.SYNC_BN[0] = True
        d = O.make_inputs(world, seed=5)
        off = d["lidar_offsets"]
        vals, offs = d["lidar_values"][off[rank]:off[rank + 1]].to(dev), (off[rank:rank + 2] - off[rank]).to(dev)
        nt = torch.nested.nested_tensor_from_jagged(vals, offs)
        out = m({"image": d["image"][rank:rank + 1].to(dev), "lidar": nt})
        g = torch.Generator().manual_seed(77)
        w_seg, w_cf = torch.randn(world, 1, 224, 224, generator=g), torch.randn(world, 4, 224, 224, generator=g)
        ((out["seg"] * w_seg[rank:rank + 1].to(dev)).sum() + (out["crossfield"] * w_cf[rank:rank + 1].to(dev)).sum()).backward()
        keys = ("encoder.proj.2.weight", "encoder.proj.2.bias", "seg_module.1.weight", "crossfield_module.1.bias", "seg_module.0.weight",
                "encoder.fusion_layer.1.weight", "encoder.proj.1.weight")
        grads = {}
        for k, p_ in m.named_parameters():
            if k in keys:
                gg = p_.grad.float().cpu()
                dist.all_reduce(gg)
                grads[k] = gg
        torch.save(dict(seg=out["seg"].detach().cpu(), cf=out["crossfield"].detach().cpu(), grads=grads,
                        rmean=dict(m.named_buffers())["seg_module.1.running_mean"].cpu()), os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_ffl_two_ranks_with_sync_batchnorm_equal_the_oracle_on_the_whole_batch():
    """configs[4], DDP / SyncBatchNorm half (model_ffl.py:161-163): the three FFL BatchNorms (proj, seg head, crossfield head) + the fusion
    and pillar BatchNorms use joint statistics: one tile per rank == the oracle's training-mode forward on both tiles; BatchNorm / conv
    gradients (summed over ranks) == autograd of the oracle."""
    import torch.multiprocessing as mp
    from oracle import p3_oracle as O
    from tests.helpers import l2_err, rel_err
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        _spawn(_worker_ffl, world, outdir)
        res = [torch.load(os.path.join(outdir, f"rank{r}.pt")) for r in range(world)]
    sd = O.make_ffl_state_dict("fusion", SMALL, seed=11)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    d = O.make_inputs(world, seed=5)
    ref, _ = O.ffl_forward(p, d["image"].double(), (d["lidar_values"], d["lidar_offsets"]), SMALL, 224, True)
    g = torch.Generator().manual_seed(77)
    w_seg, w_cf = torch.randn(world, 1, 224, 224, generator=g), torch.randn(world, 4, 224, 224, generator=g)
    ((ref["seg"] * w_seg.double()).sum() + (ref["crossfield"] * w_cf.double()).sum()).backward()
    for r in range(world):
        assert rel_err(res[r]["seg"], ref["seg"][r:r + 1].detach()) < 1e-3
        assert rel_err(res[r]["cf"], ref["crossfield"][r:r + 1].detach()) < 1e-3
    assert torch.equal(res[0]["rmean"], res[1]["rmean"])
    for k, gk in res[0]["grads"].items():
        assert l2_err(gk, p[k].grad, floor=1e-6) < 1.5e-2, (k, l2_err(gk, p[k].grad, floor=1e-6))
