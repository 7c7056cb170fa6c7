"""Training-loop sanity on the GPU: the whole stack (forward, losses, hand-written backward, flat AdamW, dropout, hipGraph-free loop)
must drive the loss down on a fixed batch, in both precision modes, and the bf16 run must track the fp32 run."""
import pytest
import torch

from oracle import p3_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _run(precision, steps, dropout):
    from pixelspointspolygons_amd import ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, train_step
    from pixelspointspolygons_amd.vision_transformer import compute_dtype
    torch.manual_seed(0)
    cfg = make_config("early_fusion_vit", precision=precision, device=DEV, vit_depth=4)
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0).train()
    sd = O.make_state_dict("fusion", dict(dim=384, depth=4, heads=6, mlp=1536, patch=8, img=224, eps=1e-6), seed=1)
    m.load_state_dict(sd, strict=True)
    if not dropout:
        m.decoder.set_dropout(0.0)
    ops.manual_seed(123, DEV)
    opt = FlatAdamW(m, lr=3e-4, weight_decay=1e-4, betas=(0.9, 0.95), compute_dtype=compute_dtype(cfg))
    inp = O.make_inputs(4, seed=21)
    batch = {"image": inp["image"].to(DEV), "lidar": (inp["lidar_values"].to(DEV), inp["lidar_offsets"].to(DEV)),
             "y": inp["y"].to(DEV), "y_perm": inp["y_perm"].to(DEV)}
    losses = []
    for _ in range(steps):
        loss, ce, bce = train_step(m, opt, batch)
        losses.append(float(loss))
    ops.DIRECT_GRAD[0] = False
    return losses


def test_overfit_fixed_batch_bf16_tracks_fp32():
    l32 = _run("fp32", 60, dropout=False)
    l16 = _run("bf16", 60, dropout=False)
    assert all(x == x for x in l32 + l16)                       # no NaN
    assert l32[-1] < 0.9 * l32[0] and l16[-1] < 0.9 * l16[0], (l32[0], l32[-1], l16[0], l16[-1])       # 6.44 -> ~5.3 in 60 steps
    assert abs(l16[0] - l32[0]) < 0.05 * l32[0]
    assert abs(l16[-1] - l32[-1]) < 0.15 * l32[0], (l32[-1], l16[-1])


def test_overfit_with_decoder_dropout():
    l = _run("bf16", 60, dropout=True)
    assert all(x == x for x in l) and min(l[-10:]) < 0.92 * l[0], (l[0], l[-10:])


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
def test_deferred_layernorm_parameter_reduces_are_bit_identical(precision):
    """ops.DEFER_PARAM_REDUCE (p3_reduce_defer / p3_reduce_flush): the LayerNorm backward launches park their dgamma / dbeta partials and ONE launch at the end of
    the backward pass adds them - the same fixed-order float64 sums: in the fp32 mode (every reduction of the step deterministic) every gradient of the train step (but the two that are summed with atomics in every mode) equals
    the immediate-reduce form bit for bit; nothing stays parked after backward()."""
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import hip, ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, pix2poly_loss
    sd = O.make_state_dict("image", seed=42)
    inp = {k: v.to("cuda") for k, v in O.make_inputs(2, seed=11).items()}

    def run(defer):
        ops.reset_process_state()
        was, ops.DEFER_PARAM_REDUCE[0] = ops.DEFER_PARAM_REDUCE[0], defer
        try:
            cfg = make_config("vit", precision=precision, device="cuda")
            m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
            m.load_state_dict(sd, strict=True)
            m.train()
            m.decoder.set_dropout(0.0)
            opt = FlatAdamW(m, compute_dtype=torch.float32)
            opt.zero_grad()
            logits, perm = m(inp["image"], None, inp["y"][:, :-1])
            pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])[0].backward()
            assert hip.reduce_pending() == 0          # LayerNorm partials AND (r05) the split-M partial tiles of the weight gradients (p3_tn_defer): all flushed
            grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
            opt.close()
            return grads
        finally:
            ops.DEFER_PARAM_REDUCE[0] = was
    g1, g0 = run(True), run(False)
    ln = [k for k in g1 if ".norm" in k or k.endswith("norm.weight") or k.endswith("norm.bias")]
    assert len(ln) >= 2 * (2 * 12 + 1 + 3 * 6)
    assert all(float(g1[k].abs().max()) > 0 for k in ln)
    # (until r05 the embedding gradient and the Sinkhorn bin score were summed with fp32 atomics in every mode; they are fixed-order sums now and compared like the rest)
    atomic = set()
    diff = {k: float((g1[k] - g0[k]).abs().max() / g0[k].abs().max().clamp_min(1e-30)) for k in g1 if k not in atomic and not torch.equal(g1[k], g0[k])}
    assert not diff, (len(diff), sorted(diff.items(), key=lambda kv: -kv[1])[:8])
    assert all(torch.allclose(g1[k], g0[k], rtol=1e-5, atol=1e-6 * float(g0[k].abs().max())) for k in atomic)


def test_a_backward_pass_that_raises_does_not_poison_the_next_step():
    """the deferred LayerNorm reduces are flushed by a final callback of the autograd engine, which a backward pass that RAISES never runs - the
    partials stay parked and the latch stays set.  FlatAdamW.zero_grad() drops them (they belong to the discarded gradients) and apply() settles whatever is
    left: the step after the failure gives bit for bit the gradients of a process that never failed."""
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import hip, ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, pix2poly_loss
    sd = O.make_state_dict("image", seed=42)
    inp = {k: v.to("cuda") for k, v in O.make_inputs(2, seed=11).items()}

    class _Boom(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x.view_as(x)

        @staticmethod
        def backward(ctx, g):
            raise RuntimeError("boom")

    def run(fail_first):
        ops.reset_process_state()
        cfg = make_config("vit", precision="fp32", device="cuda")
        m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
        m.load_state_dict(sd, strict=True)
        m.train()
        m.decoder.set_dropout(0.0)
        opt = FlatAdamW(m, compute_dtype=torch.float32)
        if fail_first:
            opt.zero_grad()
            enc = m.encoder(inp["image"])
            logits, _ = m.decoder(_Boom.apply(enc), inp["y"][:, :-1])      # the decoder's LayerNorm launches park, then the node in front of the encoder raises
            with pytest.raises(RuntimeError, match="boom"):
                logits.sum().backward()
            assert hip.reduce_pending() > 0 and ops._flush_queued[0]       # the state ADVICE r04 describes: partials parked, latch set, no callback alive ...
        opt.zero_grad()                                          # ... is gone after zero_grad
        assert hip.reduce_pending() == 0 and not ops._flush_queued[0]
        logits, perm = m(inp["image"], None, inp["y"][:, :-1])
        pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])[0].backward()
        assert hip.reduce_pending() == 0
        grads = {k: p.grad.detach().float().clone() for k, p in m.named_parameters()}
        opt.apply()
        opt.close()
        return grads
    g_fail, g_ok = run(True), run(False)
    atomic = {"bin_score", "decoder.embedding.weight"}
    bad = [k for k in g_ok if k not in atomic and not torch.equal(g_fail[k], g_ok[k])]
    assert not bad, bad[:8]


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
def test_parity_mode_train_step_is_bit_reproducible(precision):
    """The reference seeds everything and sets cudnn.deterministic (misc/shared_utils.py:120-126).  Here, with P3_DETERMINISTIC >= 1 (the default), every kernel of
    the fp32-family train step that would finish in fp32 atomics - BatchNorm sums of the pillar stem / fusion conv / ScoreNet, split-M weight gradients, bias and
    LayerNorm parameter gradients, the embedding gradient, the loss scalars, the Sinkhorn dustbin gradient - adds its workgroup partials in a fixed order in float64
    instead: two runs of the same early-fusion step from the same state give the same BITS in every output, every running statistic and every parameter gradient."""
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd import hip, ops
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    from pixelspointspolygons_amd.training import FlatAdamW, pix2poly_loss
    assert hip.DETERMINISTIC >= 1
    sd = O.make_state_dict("fusion", seed=42)
    inp = {k: v.to("cuda") for k, v in O.make_inputs(4, seed=5).items()}

    def run():
        ops.reset_process_state()
        cfg = make_config("early_fusion_vit", precision=precision, device="cuda")
        m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
        m.load_state_dict(sd, strict=True)
        m.train()
        opt = FlatAdamW(m, compute_dtype=torch.float32)       # the bench's wiring: gradients accumulated in place in the flat arena; decoder dropout ON (counter-based masks)
        ops.manual_seed(1234, "cuda")
        opt.zero_grad()
        ops.advance_rng("cuda")
        logits, perm = m(inp["image"], (inp["lidar_values"], inp["lidar_offsets"]), inp["y"][:, :-1])
        loss = pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])[0]
        loss.backward()
        torch.cuda.synchronize()
        out = {"loss": loss.detach().clone(), "logits": logits.detach().clone(), "perm": perm.detach().clone()}
        out.update({"g:" + k: p.grad.detach().clone() for k, p in m.named_parameters()})
        out.update({"b:" + k: v.detach().clone() for k, v in m.named_buffers() if v.is_floating_point()})
        opt.close()
        return out
    a, b = run(), run()
    ops.reset_process_state()
    bad = sorted(((float((a[k].float() - b[k].float()).abs().max() / a[k].float().abs().max().clamp_min(1e-30)), k) for k in a if not torch.equal(a[k], b[k])), reverse=True)
    assert not bad, (len(bad), bad[:10])


def test_parked_weight_gradients_that_share_a_target_are_not_flushed_side_by_side():
    """csrc/gemm_tn.hip p3_tn_park: the ONE flush launch adds every parked set with plain read-modify-writes from different workgroups, so a second product into an
    overlapping target (a shared weight, a second gradient into the same rows) must not be parked beside the first - it is reduced at once.  Both sums end up in the target."""
    import pixelspointspolygons_amd.hip as h
    if not h.tn_defer_arena():
        pytest.skip("weight-gradient parking is switched off (P3_TN_DEFER_MB=0)")
    g = torch.Generator().manual_seed(3)
    a1, b1 = torch.randn(4096, 256, generator=g).to(DEV), torch.randn(4096, 128, generator=g).to(DEV)
    a2, b2 = torch.randn(4096, 256, generator=g).to(DEV), torch.randn(4096, 128, generator=g).to(DEV)
    out, other = torch.zeros(256, 128, device=DEV), torch.zeros(256, 128, device=DEV)
    h.reduce_drop()
    with h.tn_parking(True):
        h.gemm_tn(a1, b1, out=out)
        n1 = h.reduce_pending()
        h.gemm_tn(a2, b2, out=out)                     # same target: reduced at once
        n2 = h.reduce_pending()
        h.gemm_tn(a2, b2, out=other)                   # another target: parked
        n3 = h.reduce_pending()
    assert (n1, n2, n3) == (1, 1, 2), (n1, n2, n3)
    h.reduce_flush()
    assert h.reduce_pending() == 0
    ref = a1.double().t() @ b1.double() + a2.double().t() @ b2.double()
    assert float((out.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    assert float((other.double() - a2.double().t() @ b2.double()).abs().max() / ref.abs().max()) < 2e-6
