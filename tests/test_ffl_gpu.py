"""FFL (frame-field) model on the HIP path vs the oracle: ViT-CNN encoders + seg / crossfield heads."""
import pytest
import torch

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _model(encoder, precision, kind, cfg_o, **kw):
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.ffl import FFLModel
    cfg = make_config(encoder, model="ffl", precision=precision, **kw)
    m = FFLModel(cfg, 0)
    sd = O.make_ffl_state_dict(kind, cfg_o, seed=11)
    m.load_state_dict({k: v.clone() for k, v in sd.items()})
    return m, sd, cfg


SMALL = dict(dim=384, depth=2, heads=6, mlp=1536, patch=8, img=224, eps=1e-6)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("fp32x3", 1e-3), ("bf16", 4e-2)])
@pytest.mark.parametrize("training", [False, True])
def test_ffl_early_fusion_forward(precision, tol, training):
    from oracle import p3_oracle as O
    m, sd, cfg = _model("early_fusion_vit_cnn", precision, "fusion", SMALL, vit_depth=2)
    B = 2 if training else 1
    d = O.make_inputs(B, seed=5)
    img, lidar_vals, lidar_offs = d["image"], d["lidar_values"], d["lidar_offsets"]
    m.train(training)
    nt = torch.nested.nested_tensor_from_jagged(lidar_vals.cuda(), lidar_offs.cuda())
    with torch.no_grad():
        out = m({"image": img.cuda(), "lidar": nt})
        ref, feats = O.ffl_forward({k: v.clone() for k, v in sd.items()}, img, (lidar_vals, lidar_offs), SMALL, 224, training)
    assert out["seg"].shape == (B, 1, 224, 224) and out["crossfield"].shape == (B, 4, 224, 224)
    if precision == "bf16":
        # bf16 storage through three batch-normalised 3x3 convs: single pixels near a ReLU / tanh knee move by ~0.1; the bound that
        # means something is the L2-relative error of the whole map (max-abs checked loosely)
        l2 = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
        assert l2(out["seg"].cpu(), ref["seg"]) < tol / 2 and l2(out["crossfield"].cpu(), ref["crossfield"]) < tol / 2
        assert rel_err(out["crossfield"].cpu(), ref["crossfield"]) < 0.2
    else:
        assert rel_err(out["seg"].cpu(), ref["seg"]) < tol
        assert rel_err(out["crossfield"].cpu(), ref["crossfield"]) < tol
    # the encoder's own forward (NCHW features) — the reference's EarlyFusionViTCNN.forward contract
    if not training:
        with torch.no_grad():
            f = m.encoder(img.cuda(), nt)
        assert f.shape == (B, 256, 224, 224)
        assert rel_err(f.cpu(), feats) < tol


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
@pytest.mark.parametrize("encoder,kind", [("vit_cnn", "image"), ("pointpillars_vit_cnn", "lidar")])
def test_ffl_single_modality(encoder, kind, precision):
    from oracle import p3_oracle as O
    m, sd, cfg = _model(encoder, precision, kind, SMALL, vit_depth=2)
    d = O.make_inputs(1, seed=6)
    img, lidar_vals, lidar_offs = d["image"], d["lidar_values"], d["lidar_offsets"]
    nt = torch.nested.nested_tensor_from_jagged(lidar_vals.cuda(), lidar_offs.cuda())
    m.eval()
    with torch.no_grad():
        if kind == "image":
            out = m({"image": img.cuda()})
            ref, _ = O.ffl_forward(sd, img=img, cfg=SMALL)
        else:
            out = m({"lidar": nt})
            ref, _ = O.ffl_forward(sd, lidar=(lidar_vals, lidar_offs), cfg=SMALL)
    assert rel_err(out["seg"].cpu(), ref["seg"]) < 1e-3
    assert rel_err(out["crossfield"].cpu(), ref["crossfield"]) < 1e-3


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
@pytest.mark.parametrize("training", [True, False])
def test_cnn_encoder_standalone_forward_is_differentiable(training, precision):
    """ViTCNN.forward (vit_cnn.py:45-57) on its own: NCHW features and the gradients of every encoder parameter + the input image
    for a random linear functional, vs float64 autograd of the oracle (encoder tokens -> vitcnn_tail)."""
    from oracle import p3_oracle as O
    from helpers import l2_err
    m, sd, cfg = _model("vit_cnn", precision, "image", SMALL, vit_depth=2)
    enc = m.encoder.train(training)
    img = O.make_inputs(1, seed=12)["image"]
    gen = torch.Generator().manual_seed(5)
    g = torch.randn(1, 256, 224, 224, generator=gen)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    x = O.patch_embed(img.double(), p, "encoder.vit.patch_embed.", SMALL["patch"]).flatten(2).transpose(1, 2)
    tok = O.vit_blocks(x, p, "encoder.vit.", SMALL["depth"], SMALL["heads"], SMALL["eps"])
    ref = O.vitcnn_tail(tok, p, "encoder.", 224, training)
    (ref * g.double()).sum().backward()
    xin = img.cuda().requires_grad_(True)
    out = enc(xin)
    assert out.shape == (1, 256, 224, 224) and out.dtype == torch.float32
    assert rel_err(out.detach().cpu(), ref.detach()) < 1e-3
    (out * g.cuda()).sum().backward()
    gnorm = max(float(v.grad.norm()) for k, v in p.items() if k.startswith("encoder.") and v.is_floating_point() and v.requires_grad and v.grad is not None)
    bad = {}
    for k, prm in enc.named_parameters():
        r = p["encoder." + k].grad
        if r is None:
            continue
        assert prm.grad is not None, k
        e = l2_err(prm.grad.float().cpu(), r, floor=1e-3 * gnorm)
        if not e < 1.5e-2:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


@pytest.mark.parametrize("precision", ["fp32", "fp32x3"])
@pytest.mark.parametrize("training", [True, False])
def test_ffl_backward_vs_oracle_autograd(training, precision):
    """FFL training path (a-14): gradients of every parameter (heads, BatchNorms, 3x3 convs, proj, fusion stem, ViT) for a random
    linear functional of (seg, crossfield) vs float64 autograd of the oracle; train- and eval-mode BatchNorm."""
    from oracle import p3_oracle as O
    from helpers import l2_err
    m, sd, cfg = _model("early_fusion_vit_cnn", precision, "fusion", SMALL, vit_depth=2)
    B = 1
    d = O.make_inputs(B, seed=9)
    img, lv, lo = d["image"], d["lidar_values"], d["lidar_offsets"]
    gen = torch.Generator().manual_seed(4)
    g1, g2 = torch.randn(B, 1, 224, 224, generator=gen), torch.randn(B, 4, 224, 224, generator=gen)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items()}
    ref, _ = O.ffl_forward(p, img.double(), (lv, lo), SMALL, 224, training)
    ((ref["seg"] * g1.double()).sum() + (ref["crossfield"] * g2.double()).sum()).backward()
    m.train(training)
    nt = torch.nested.nested_tensor_from_jagged(lv.cuda(), lo.cuda())
    out = m({"image": img.cuda(), "lidar": nt})
    assert rel_err(out["seg"].detach().cpu(), ref["seg"].detach()) < 1e-3
    ((out["seg"] * g1.cuda()).sum() + (out["crossfield"] * g2.cuda()).sum()).backward()
    gnorm = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad and v.grad is not None)
    bad = {}
    for k, prm in m.named_parameters():
        r = p[k].grad
        if r is None:
            continue
        assert prm.grad is not None, k
        e = l2_err(prm.grad.float().cpu(), r, floor=1e-3 * gnorm)
        if not e < 1.5e-2:
            bad[k] = e
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:10]


FULL = dict(dim=384, depth=12, heads=6, mlp=1536, patch=8, img=224, eps=1e-6)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "bf16"])
def test_ffl_full_batch_is_batch_independent_and_matches_the_oracle(precision):
    """BASELINE configs[4] at its REAL size (model_ffl.py:28-104 over early_fusion_vit_cnn.py:87-104): ViT depth 12, 64 tiles of 224 x 224 +
    3 k points.  Eval mode: tile b of the batch of 64 == the same tile run alone, bit for bit (row-independent implicit-GEMM convs, per-tile
    upsample / pillar sort), and two of the 64 tiles are checked against the oracle."""
    from oracle import p3_oracle as O
    m, sd, cfg = _model("early_fusion_vit_cnn", precision, "fusion", FULL, vit_depth=12, batch_size=64)
    m.eval()
    B = 64
    inp = O.make_inputs(B, seed=1234)
    off = inp["lidar_offsets"]
    img, lv, lo = inp["image"].cuda(), inp["lidar_values"].cuda(), inp["lidar_offsets"].cuda()
    with torch.no_grad():
        out = m({"image": img, "lidar": torch.nested.nested_tensor_from_jagged(lv, lo)})
        seg, cf = out["seg"], out["crossfield"]
        assert seg.shape == (B, 1, 224, 224) and cf.shape == (B, 4, 224, 224)
        assert torch.isfinite(seg).all() and torch.isfinite(cf).all()
        assert float(seg.min()) >= 0.0 and float(seg.max()) <= 1.0 and float(cf.abs().max()) <= 2.0      # sigmoid / 2 * tanh ranges (model_ffl.py:87-93)
        for b in (0, 29, 63):
            vals = lv[off[b]:off[b + 1]].contiguous()
            offs = torch.tensor([0, int(off[b + 1] - off[b])], device="cuda")
            o1 = m({"image": img[b:b + 1], "lidar": torch.nested.nested_tensor_from_jagged(vals, offs)})
            assert torch.equal(o1["seg"][0], seg[b]) and torch.equal(o1["crossfield"][0], cf[b]), b
        for b in (29, 63):
            l0, l1 = int(off[b]), int(off[b + 1])
            ref, _ = O.ffl_forward({k: v.clone() for k, v in sd.items()}, inp["image"][b:b + 1],
                                   (inp["lidar_values"][l0:l1], torch.tensor([0, l1 - l0])), FULL, 224, False)
            if precision != "bf16":
                assert rel_err(seg[b:b + 1].float().cpu(), ref["seg"]) < 1e-3
                assert rel_err(cf[b:b + 1].float().cpu(), ref["crossfield"]) < 1e-3
            else:
                l2 = lambda a, r: float((a.double() - r.double()).norm() / r.double().norm())
                assert l2(seg[b:b + 1].float().cpu(), ref["seg"]) < 2e-2 and l2(cf[b:b + 1].float().cpu(), ref["crossfield"]) < 2e-2
                assert rel_err(cf[b:b + 1].float().cpu(), ref["crossfield"]) < 0.2
