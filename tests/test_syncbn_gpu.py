"""SyncBatchNorm + data-parallel gradient averaging (model_pix2poly.py:326-328) on the HIP path: two ranks, each with
half of a batch, must reproduce the oracle run on the WHOLE batch (joint BatchNorm statistics in all 9 BatchNorm sites, forward and
backward).  The two ranks share the box's single GPU and talk over gloo (ops.sync_stats reduces through the host for gloo)."""
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


def _worker(rank, world, port, outdir, precision):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import pix2poly_loss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = "cuda:0"
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


@pytest.mark.parametrize("precision", ["fp32"])
def test_two_ranks_with_sync_batchnorm_equal_the_oracle_on_the_whole_batch(precision):
    import torch.multiprocessing as mp
    from oracle import p3_oracle as O
    from tests.helpers import l2_err, rel_err
    world = 2
    with tempfile.TemporaryDirectory() as outdir:
        mp.spawn(_worker, args=(world, _free_port(), outdir, precision), nprocs=world, join=True)
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
        assert rel_err(res[r]["perm"], perm[2 * r:2 * r + 2].detach()) < 5e-3
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
