"""Pins the oracle (oracle/p3_oracle.py) against fixtures emitted by the REFERENCE's own modules
(tests/golden/make_golden.py, run in the build container).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import load_golden, rel_err

TOL = 2e-5   # fp32 CPU vs fp32 CPU, different op order


def _full_sd(seed):
    return O.make_state_dict("image", dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=224, eps=1e-6),
                             seed=seed, n_vertices=192, dec_dim=256, dec_layers=6)


def test_decoder_small_forward_and_predict():
    d, w = load_golden("decoder_small.npz")
    logits, feats = O.decoder_forward(d["enc"], d["y"], w, layers=2)
    assert rel_err(logits, d["logits"]) < TOL and rel_err(feats, d["feats"]) < TOL
    pl, pf = O.decoder_predict(d["enc"], d["y"][:, :5], w, max_len=22, layers=2)
    assert rel_err(pl, d["pred_logits"]) < TOL and rel_err(pf, d["pred_feats"]) < TOL


def test_float_padding_mask_is_additive_plus_one():
    """(tgt==PAD).float() acts as +1.0 bias; a boolean mask would give different rows >= first PAD."""
    d, w = load_golden("decoder_small.npz")
    y = d["y"]
    assert (y == O.PAD).any()
    logits, _ = O.decoder_forward(d["enc"], y, w, layers=2)
    assert rel_err(logits, d["logits"]) < TOL
    causal, kpm = O.create_mask(y)
    assert set(kpm.unique().tolist()) <= {0.0, 1.0} and torch.isinf(causal).any()


def test_decoder_full_shape():
    d, _ = load_golden("decoder_full.npz")
    sd = _full_sd(42)
    logits, feats = O.decoder_forward(d["enc"], d["y"], sd)
    assert rel_err(logits, d["logits"]) < TOL and rel_err(feats, d["feats"]) < TOL
    pl, _ = O.decoder_predict(d["enc"], d["y"][:, :5], sd)
    assert rel_err(pl, d["pred_logits"]) < TOL


def test_greedy_tokens_bit_exact():
    d, _ = load_golden("greedy_small.npz")
    _, w = load_golden("decoder_small.npz")
    enc = d["enc"]
    preds = torch.full((2, 1), O.BOS, dtype=torch.long)
    for _ in range(21):
        lg, feats = O.decoder_predict(enc, preds, w, max_len=22, layers=2)
        preds = torch.cat([preds, torch.softmax(lg, -1).argmax(-1, keepdim=True)], 1)
    assert torch.equal(preds, d["tokens"])
    assert rel_err(feats, d["feats"]) < TOL


def test_scorenet_small_eval_and_train():
    d, w = load_golden("scorenet_small.npz")
    for s in ("scorenet1.", "scorenet2."):
        out = O.scorenet(d["feats"], w, s, n_vertices=10, training=False)
        assert rel_err(out, d[s + "eval"]) < TOL
        w2 = {k: v.clone() for k, v in w.items()}
        out = O.scorenet(d["feats"], w2, s, n_vertices=10, training=True)
        assert rel_err(out, d[s + "train"]) < 1e-4
        assert rel_err(w2[s + "bn1.running_mean"], d[s + "rm1"]) < 1e-5
        assert rel_err(w2[s + "bn1.running_var"], d[s + "rv1"]) < 1e-5


def test_scorenet_full_eval():
    d, _ = load_golden("scorenet_full.npz")
    sd = _full_sd(42)
    out = O.scorenet(d["feats"], sd, "scorenet1.", training=False)
    assert rel_err(out, d["scorenet1.eval"]) < TOL


def test_sinkhorn():
    d, _ = load_golden("sinkhorn.npz")
    one = torch.tensor(1.0)
    assert rel_err(O.log_optimal_transport(d["scores_small"], one, 100), d["lot_small"]) < TOL
    assert rel_err(O.log_optimal_transport(d["scores_full"], one, 100), d["lot_full"]) < TOL
    assert rel_err(O.log_optimal_transport(d["scores_small"], torch.tensor(0.3), 3), d["lot_small_it3"]) < TOL


def test_encoder_decoder_glue_small():
    """Reference EncoderDecoder.forward + EarlyFusionViT.forward glue (hybrid oracle)."""
    d, w = load_golden("encdec_small.npz")
    vc = dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=32, eps=1e-6)
    enc = O.encoder_fusion(d["image"], d["lidar_values"], d["lidar_offsets"], w, vc, out_dim=64,
                           grid=(4, 4), max_points=8, max_voxels=16)
    logits, feats = O.decoder_forward(enc, d["y"][:, :-1], w, layers=2)
    s = O.scorenet(feats, w, "scorenet1.", 10) + O.scorenet(feats, w, "scorenet2.", 10).transpose(1, 2)
    perm = torch.softmax(O.log_optimal_transport(s, w["bin_score"], 100)[:, :10, :10], -1)
    assert rel_err(logits, d["seq_pred"]) < TOL
    assert rel_err(perm, d["perm_mat"]) < 1e-4


def test_vit_body_matches_independent_implementation():
    d, w = load_golden("vit_hf_small.npz")
    vc = dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=32, eps=1e-6)
    x = O.patch_embed(d["image"], w, "encoder.vit.patch_embed.", 8).flatten(2).transpose(1, 2)
    tok = O.vit_blocks(x, w, "encoder.vit.", 2, 2, 1e-6)
    assert rel_err(tok, d["tokens"]) < TOL


def test_vit_s8_full_shape_matches_independent_implementation():
    d, _ = load_golden("vit_hf_s8.npz")
    sd = O.make_state_dict("image", O.VIT_S8, seed=42)
    wsum = float(sum(v.double().sum() for v in sd.values() if v.is_floating_point()))
    assert abs(wsum - float(d["wsum"][0])) < 1e-6 * abs(wsum) + 1e-6, "torch RNG drifted: regenerate fixtures"
    img = d["image"].float()
    x = O.patch_embed(img, sd, "encoder.vit.patch_embed.", 8).flatten(2).transpose(1, 2)
    tok = O.vit_blocks(x, sd, "encoder.vit.", 12, 6, 1e-6)
    assert rel_err(tok[:, ::8, :], d["tokens"]) < 5e-5


def test_tokenizer_constants_and_roundtrip():
    d, _ = load_golden("tokenizer.npz")
    assert d["consts"].tolist() == [O.BOS, O.EOS, O.PAD, O.VOCAB, O.MAX_LEN, O.MAX_LEN - 1]
    coords = d["coords"].numpy()
    q = np.round(coords / 224 * 223).astype(int)
    assert d["tokens"].tolist() == [O.BOS] + q.reshape(-1).tolist() + [O.EOS]


def test_pool_pairs_pattern():
    """AdaptiveAvgPool1d(384->256) windows {0,1},{1,2},{3,4},{4,5},..."""
    x = torch.arange(384.0).view(1, 1, 384)
    y = O.pool_channels(x, 256)[0, 0]
    exp = torch.tensor([(3 * (c // 2) + (c % 2)) + 0.5 for c in range(256)])
    assert torch.equal(y, exp)


def test_demo_tile_fixture_is_consistent_with_the_oracle():
    """configs[0] fixture: planted decoder tensors + seeded model -> the oracle's teacher-forced argmax reproduces the stored tokens up
    to the EOS (one full decoder pass: cheap), the stored permutation is the scipy assignment of the stored scores."""
    import os
    from tests.helpers import GOLD
    fx = np.load(os.path.join(GOLD, "demo_tile.npz"))
    sd = O.make_state_dict("image", O.VIT_S8, seed=42)
    for k in fx.files:
        if k.startswith("planted."):
            sd[k[len("planted."):]] = torch.from_numpy(fx[k])
    img = torch.from_numpy(fx["image_u8"]).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    toks = torch.from_numpy(fx["tokens"])
    with torch.no_grad():
        enc = O.encoder_vit(img, sd, cfg=O.VIT_S8)
        logits, _ = O.decoder_forward(enc, toks[:, :-1], sd)
    eos = int((toks[0] == O.EOS).nonzero()[0])
    assert torch.equal(logits[0, :eos].argmax(-1), toks[0, 1:eos + 1])        # causal decoder: teacher forcing == the greedy loop's choices
    assert torch.equal(O.scores_to_permutations(torch.from_numpy(fx["scores"])), torch.from_numpy(fx["perm"]).float())
    assert fx["image_u8"].shape == (224, 224, 3) and fx["poly_len"].sum() == len(fx["poly_flat"])


def test_hisup_head_set_matches_the_reference_module():
    """`EncoderDecoder.forward_common` of the HiSup model (model_hisup.py:205-226) on a fixed feature map - eval mode
    (running statistics) and one training-mode forward (batch statistics + the running-statistics update), golden from the reference's
    own module (tests/golden/make_hisup_heads_golden.py)."""
    d, w = load_golden("hisup_heads.npz")
    out = O.hisup_heads(d["features"], w, training=False)
    assert set(out) == {"joff", "jloc", "mask", "afm", "remask"}
    for k, v in out.items():
        assert v.shape == d["eval." + k].shape and rel_err(v, d["eval." + k]) < 1e-5, k
    w2 = {k: v.clone() for k, v in w.items()}
    out = O.hisup_heads(d["features"], w2, training=True)
    for k, v in out.items():
        assert rel_err(v, d["train." + k]) < 1e-4, k
    n = 0
    for k, v in d.items():
        if k.startswith("after."):
            name = k[len("after."):]
            if "num_batches" in name:
                assert int(w2[name]) == int(v)
            else:
                assert rel_err(w2[name], v) < 1e-5, name
            n += 1
    assert n == 3 * 17                     # 17 BatchNorm2d sites: 9 in the three towers, 2 ECA, 3 + 3 in refuse / final conv



@pytest.mark.parametrize("train,transpose", [(True, False), (True, True), (False, False)])
def test_scorenet_staged_backward_equals_autograd_of_the_dense_oracle(train, transpose):
    """`scorenet_staged` (float64, analytic backward in the separable U_i + V_j staging, used by the GPU backward test to replay ReLU
    decisions at the kink) is the same function and the same gradient as float64 autograd of the dense restatement `scorenet`, which is
    pinned against the reference's own ScoreNet (scorenet_*.npz)."""
    N, B = 24, 3
    sd = O.make_state_dict("image", dict(dim=64, depth=1, heads=2, mlp=128, patch=8, img=32, eps=1e-6), seed=9, n_vertices=N)
    gen = torch.Generator().manual_seed(4)
    feats, g = torch.randn(B, 2 * N + 1, 256, generator=gen), torch.randn(B, N, N, generator=gen)
    p = {k: (v.double().requires_grad_(True) if v.is_floating_point() and "running" not in k else (v.double() if v.is_floating_point() else v.clone()))
         for k, v in sd.items() if k.startswith("scorenet1.")}
    f = feats.double().requires_grad_(True)
    s = O.scorenet(f, p, "scorenet1.", n_vertices=N, training=train)
    (s.transpose(1, 2) if transpose else s).backward(g.double())
    out, grads, dfeats, zs = O.scorenet_staged(feats, g, sd, "scorenet1.", n_vertices=N, training=train, transpose=transpose)
    assert float((out - (s.transpose(1, 2) if transpose else s)).abs().max()) < 1e-12
    gn = max(float(v.grad.norm()) for v in p.values() if v.is_floating_point() and v.requires_grad)
    for k, v in grads.items():
        assert float((v - p["scorenet1." + k].grad).norm()) < 1e-11 * gn, k
    assert float((dfeats - f.grad).norm()) < 1e-11 * float(f.grad.norm())
    assert all(z.shape[0] == B * N * N for z in zs)
