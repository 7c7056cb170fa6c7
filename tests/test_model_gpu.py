"""End-to-end parity of the HIP path behind the reference's nn.Module API (GPU box only).

Parity modes - 'fp32' (exact fp32 MFMA) and 'fp32x3' (fp32 storage, every product as bf16 x 3: the bench's headline mode) - run the SAME tests: logits /
features / scores / perm within 1e-3 rel of the oracle and of the golden fixtures emitted by the reference's own modules; integer outputs (greedy tokens) bit-exact.
Throughput mode (precision='bf16'): same checks at the documented bf16 tolerance.
"""
import pytest
import torch

from oracle import p3_oracle as O
from tests.helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
TOL32 = 1e-3      # BASELINE.json north_star: "fp logits within 1e-3 rel"
TOL16 = 6e-2      # bf16 storage through 12 + 6 layers (documented in DESIGN.md)
PARITY_MODES = ["fp32", "fp32x3"]     # both are held to the north star's tolerance


def perm_tol(tol):
    """the permutation matrix is held to the SAME 1e-3 as the logits in the fp32 parity mode (measured 2e-6); the bf16 mode keeps 5 x its tolerance"""
    return tol if tol <= TOL32 else tol * 5


def _model(kind, precision, sd=None, **kw):
    from pixelspointspolygons_amd.config import make_config
    from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
    name = {"fusion": "early_fusion_vit", "image": "vit", "lidar": "pointpillars_vit"}[kind]
    cfg = make_config(name, precision=precision, device=DEV, **kw)
    tk = Tokenizer(cfg)
    m = Pix2PolyModel(cfg, tk.vocab_size, 0)
    if sd is not None:
        missing = m.load_state_dict({k: v for k, v in sd.items()}, strict=True)
    return m.eval(), cfg


def _to_dev(inp):
    return {k: v.to(DEV) for k, v in inp.items()}


@pytest.fixture(autouse=True)
def _no_precision_scope_leaks():
    """the product precision is a scope of the model that launches (hip.scope_module / @precision_scoped), not a process setting: no test may leave one open"""
    import pixelspointspolygons_amd.hip as hip
    assert not hip.split_now()
    yield
    assert not hip.split_now()


@pytest.mark.parametrize("precision,tol", [("fp32", TOL32), ("fp32x3", TOL32), ("bf16", TOL16)])
@pytest.mark.parametrize("kind", ["fusion", "image", "lidar"])
def test_pix2poly_forward_eval_vs_oracle(kind, precision, tol):
    sd = O.make_state_dict(kind, seed=42)
    m, cfg = _model(kind, precision, sd)
    inp = O.make_inputs(2, seed=1234)
    y = inp["y"][:, :-1]
    img = inp["image"] if kind != "lidar" else None
    lidar = (inp["lidar_values"], inp["lidar_offsets"]) if kind != "image" else None
    with torch.no_grad():
        ref_logits, ref_perm = O.pix2poly_forward({k: v.clone() for k, v in sd.items()}, y, img, lidar)
        d = _to_dev(inp)
        lj = torch.nested.nested_tensor_from_jagged(d["lidar_values"], d["lidar_offsets"]) if lidar is not None else None
        logits, perm = m(d["image"] if img is not None else None, lj, d["y"][:, :-1])
    assert logits.shape == ref_logits.shape and perm.shape == ref_perm.shape
    print(f"[{kind}-{precision}] logits {rel_err(logits.float().cpu(), ref_logits):.3e} perm {rel_err(perm.float().cpu(), ref_perm):.3e}")
    assert rel_err(logits.float().cpu(), ref_logits) < tol
    assert rel_err(perm.float().cpu(), ref_perm) < perm_tol(tol)
    if precision in ("fp32", "fp32x3"):   # bit-exact token indices
        assert torch.equal(logits.float().cpu().argmax(-1), ref_logits.argmax(-1))


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_encoder_features_and_pillar_canvas_fp32(precision):
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd)
    inp = O.make_inputs(3, seed=7, n_points=3000)
    with torch.no_grad():
        canvas_ref = O.pillar_stem(inp["lidar_values"], inp["lidar_offsets"], sd, "encoder.lidar_embed.")
        enc_ref = O.encoder_fusion(inp["image"], inp["lidar_values"], inp["lidar_offsets"], sd)
        d = _to_dev(inp)
        canvas = m.encoder.lidar_embed((d["lidar_values"], d["lidar_offsets"]), return_flattened=False)
        enc = m.encoder(d["image"], (d["lidar_values"], d["lidar_offsets"]))
    assert rel_err(canvas.float().cpu(), canvas_ref) < 1e-4
    # empty pillars are exact zeros in both
    assert torch.equal(canvas.float().cpu() == 0, canvas_ref == 0)
    assert rel_err(enc.float().cpu(), enc_ref) < TOL32


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_pillar_edge_cases_fp32(precision):
    """dense cloud (cap of 64 points hit: lowest indices kept), points on the range boundary, an empty sample."""
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd)
    g = torch.Generator().manual_seed(3)
    dense = torch.rand(20000, 3, generator=g) * torch.tensor([60.0, 60.0, 99.0])          # ~400 pts / pillar
    edge = torch.tensor([[224.0, 10.0, 5.0], [10.0, 224.0, 5.0], [100.0, 100.0, 100.0], [100.5, 100.5, 50.0],
                         [-0.1, 5.0, 5.0], [5.0, 5.0, 100.1], [223.99, 223.99, 99.99], [0.0, 0.0, 0.0]])
    vals = torch.cat([dense, edge, torch.rand(100, 3, generator=g) * 200])
    offs = torch.tensor([0, 20000, 20008, 20008, 20108])                                   # sample 2 is empty
    with torch.no_grad():
        ref = O.pillar_stem(vals, offs, sd, "encoder.lidar_embed.")
        out = m.encoder.lidar_embed((vals.to(DEV), offs.to(DEV)), return_flattened=False)
    assert rel_err(out.float().cpu(), ref) < 1e-4
    assert torch.equal(out.float().cpu() == 0, ref == 0)


@pytest.mark.parametrize("precision,tol", [("fp32", TOL32), ("fp32x3", TOL32), ("bf16", TOL16)])
def test_forward_train_mode_batchnorm_statistics(precision, tol):
    """train(): BN batch statistics (PFN x2, fusion, ScoreNet x6) + running-stat updates; dropout p = 0 for parity."""
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd)
    m.train()
    m.decoder.set_dropout(0.0)
    inp = O.make_inputs(2, seed=99)
    sd_ref = {k: v.clone() for k, v in sd.items()}
    with torch.no_grad():
        ref_logits, ref_perm = O.pix2poly_forward(sd_ref, inp["y"][:, :-1], inp["image"], (inp["lidar_values"], inp["lidar_offsets"]),
                                                  training=True)
        d = _to_dev(inp)
        logits, perm = m(d["image"], (d["lidar_values"], d["lidar_offsets"]), d["y"][:, :-1])
    assert rel_err(logits.float().cpu(), ref_logits) < tol
    assert rel_err(perm.float().cpu(), ref_perm) < perm_tol(tol)
    new = m.state_dict()
    for k in ("encoder.fusion_layer.1.running_mean", "encoder.fusion_layer.1.running_var", "scorenet1.bn2.running_var",
              "encoder.lidar_embed.voxel_encoder.pfn_layers.1.norm.running_mean", "scorenet2.bn3.running_mean"):
        assert rel_err(new[k].cpu(), sd_ref[k]) < (1e-3 if precision != "bf16" else 5e-2), k
    assert int(new["scorenet1.bn1.num_batches_tracked"]) == 1


# ---------------------------------------------------------------- against the reference's own outputs (golden fixtures)
def _decoder_from(sd, layers, nv, enc_len, precision="fp32"):
    from pixelspointspolygons_amd.pix2poly import Decoder
    dec = Decoder(vocab_size=O.VOCAB, encoder_len=enc_len, dim=256, num_heads=8, num_layers=layers, max_len=2 * nv + 2, pad_idx=O.PAD)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in sd.items() if k.startswith("decoder.")}, strict=True)
    dec.cd = torch.float32 if precision != "bf16" else torch.bfloat16
    from pixelspointspolygons_amd import hip
    hip.scope_module(dec, precision == "fp32x3", ("predict", "generate_cached"))      # what EncoderDecoder.__init__ does for its decoder
    return dec.to(DEV).eval()


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_decoder_full_shape_vs_reference_golden(precision):
    d, _ = load_golden("decoder_full.npz")
    sd = O.make_state_dict("image", dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=224, eps=1e-6), seed=42)
    dec = _decoder_from(sd, 6, 192, 784, precision)
    with torch.no_grad():
        logits, feats = dec(d["enc"].to(DEV), d["y"].to(DEV))
        pl, pf = dec.predict(d["enc"].to(DEV), d["y"][:, :5].to(DEV))
    assert rel_err(logits.cpu(), d["logits"]) < TOL32 and rel_err(feats.cpu(), d["feats"]) < TOL32
    assert rel_err(pl.cpu(), d["pred_logits"]) < TOL32 and rel_err(pf.cpu(), d["pred_feats"]) < TOL32


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_greedy_decode_tokens_bit_exact_vs_reference_golden(precision):
    d, _ = load_golden("greedy_d256.npz")
    sd = O.make_state_dict("image", dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=32, eps=1e-6), seed=77, n_vertices=10,
                           dec_dim=256, dec_layers=2)
    wsum = float(sum(v.double().sum() for k, v in sd.items() if k.startswith("decoder.")))
    assert abs(wsum - float(d["wsum"][0])) < 1e-6 * abs(wsum) + 1e-6
    dec = _decoder_from(sd, 2, 10, 16, precision)
    from pixelspointspolygons_amd import hip
    enc = d["enc"].to(DEV)
    preds = torch.full((3, 1), O.BOS, dtype=torch.long, device=DEV)
    with torch.no_grad():
        for _ in range(21):
            lg, feats = dec.predict(enc, preds)
            preds = torch.cat([preds, hip.argmax(lg).view(-1, 1)], 1)
        logits, ff = dec(enc, d["y"].to(DEV))
    assert torch.equal(preds.cpu(), d["tokens"])
    assert rel_err(feats.cpu(), d["feats"]) < TOL32
    assert rel_err(logits.cpu(), d["logits"]) < TOL32


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_scorenet_full_vs_reference_golden(precision):
    from pixelspointspolygons_amd.pix2poly import ScoreNet
    from pixelspointspolygons_amd import hip
    d, _ = load_golden("scorenet_full.npz")
    sd = O.make_state_dict("image", dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=224, eps=1e-6), seed=42)
    for s in ("scorenet1.", "scorenet2."):
        for mode in ("eval", "train"):
            net = ScoreNet(192, in_channels=512)
            net.load_state_dict({k[len(s):]: v for k, v in sd.items() if k.startswith(s)}, strict=True)
            net.cd = torch.float32
            hip.scope_module(net, precision == "fp32x3", ("scores_into",))
            net = net.to(DEV).train(mode == "train")
            with torch.no_grad():
                out = net(d["feats"].to(DEV))
            assert rel_err(out.cpu(), d[s + mode]) < TOL32, (s, mode)
            if mode == "train":
                assert rel_err(net.bn1.running_mean.cpu(), d[s + "rm1"]) < 1e-4
                assert rel_err(net.bn1.running_var.cpu(), d[s + "rv1"]) < 1e-4


def test_sinkhorn_vs_reference_golden():
    from pixelspointspolygons_amd.pix2poly import log_optimal_transport
    d, _ = load_golden("sinkhorn.npz")
    one = torch.tensor(1.0, device=DEV)
    assert rel_err(log_optimal_transport(d["scores_small"].to(DEV), one, 100).cpu(), d["lot_small"]) < 1e-5
    assert rel_err(log_optimal_transport(d["scores_full"].to(DEV), one, 100).cpu(), d["lot_full"]) < 1e-5
    assert rel_err(log_optimal_transport(d["scores_small"].to(DEV), torch.tensor(0.3, device=DEV), 3).cpu(), d["lot_small_it3"]) < 1e-5
    from pixelspointspolygons_amd import ops
    perm = ops.sinkhorn_softmax(d["scores_full"].to(DEV), one, 100).cpu()
    assert rel_err(perm, torch.softmax(d["lot_full"][:, :192, :192], -1)) < 1e-5


def test_state_dict_contract_matches_reference_key_list():
    """parameter / buffer names and shapes equal the reference's (captured in the oracle's key list)."""
    sd = O.make_state_dict("fusion", seed=1)
    m, _ = _model("fusion", "bf16")
    mine = m.state_dict()
    assert set(mine) == set(sd)
    assert all(tuple(mine[k].shape) == tuple(sd[k].shape) for k in sd)


@pytest.mark.parametrize("precision,tol", [("fp32", TOL32), ("fp32x3", TOL32), ("bf16", TOL16)])
def test_pix2poly_vit_b16_forward_vs_oracle(precision, tol):
    """BASELINE.json configs[1]: image-only ViT-B/16 (dim 768, 12 heads, 196 patches -> cross-attention over 196 keys)."""
    sd = O.make_state_dict("image", O.VIT_B16, seed=17)
    m, cfg = _model("image", precision, sd, patch_size=16, patch_feature_dim=768, vit_heads=12)
    inp = O.make_inputs(2, seed=31)
    with torch.no_grad():
        ref_logits, ref_perm = O.pix2poly_forward({k: v.clone() for k, v in sd.items()}, inp["y"][:, :-1], inp["image"], None, cfg=O.VIT_B16)
        d = _to_dev(inp)
        logits, perm = m(d["image"], None, d["y"][:, :-1])
    assert rel_err(logits.float().cpu(), ref_logits) < tol
    assert rel_err(perm.float().cpu(), ref_perm) < perm_tol(tol)
    if precision != "bf16":
        assert torch.equal(logits.float().cpu().argmax(-1), ref_logits.argmax(-1))


def test_embed_tokens_rejects_sequences_longer_than_the_positional_table():
    from pixelspointspolygons_amd.hip import P3Error
    m, cfg = _model("image", "fp32", None, max_num_vertices=16)
    y = torch.zeros(1, 385, dtype=torch.long, device=DEV)
    with pytest.raises(P3Error):
        with torch.no_grad():
            m(torch.rand(1, 3, 224, 224, device=DEV), None, y)


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_kv_cached_decode_is_bit_identical_to_the_full_rerun_and_to_the_reference_golden(precision):
    """incremental greedy decode over KV caches == the reference's loop of full `predict` passes (golden tokens)."""
    d, _ = load_golden("greedy_d256.npz")
    sd = O.make_state_dict("image", dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=32, eps=1e-6), seed=77, n_vertices=10,
                           dec_dim=256, dec_layers=2)
    dec = _decoder_from(sd, 2, 10, 16, precision).eval()
    from pixelspointspolygons_amd import hip
    enc = d["enc"].to(DEV)
    preds = torch.full((3, 1), O.BOS, dtype=torch.long, device=DEV)
    with torch.no_grad():
        for _ in range(21):
            lg, feats = dec.predict(enc, preds)
            preds = torch.cat([preds, hip.argmax(lg).view(-1, 1)], 1)
        dec.fused_decode = False                                     # the launch chain: same arithmetic per (position, channel) as predict
        toks, cfeats = dec.generate_cached(enc, 21, O.BOS)
        dec.fused_decode, dec._decode_state = True, None             # r03: one p3_decode_layer launch per layer in fp32 too
        ftoks, ffeats = dec.generate_cached(enc, 21, O.BOS)
    assert torch.equal(toks.cpu(), d["tokens"])                      # the reference's own greedy sequence
    assert torch.equal(toks, preds)
    assert torch.equal(cfeats, feats[:, :21])                        # bit-identical features (fp32 mode, launch chain)
    assert torch.equal(ftoks.cpu(), d["tokens"])                     # fused fp32 layer: the same tokens as the reference ...
    # ... features to fp32 rounding (another summation order inside the dot products; the fused layer's own dot products are exact fp32 in both parity modes)
    assert rel_err(ffeats.cpu(), cfeats.cpu()) < (2e-5 if precision == "fp32" else 1e-4)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "bf16"])
def test_kv_cached_generate_full_model(precision):
    """Whole early-fusion model, full 385-step decode: cached == literal loop on a prefix; timing of both printed."""
    import time
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd)
    inp = O.make_inputs(2, seed=3)
    d = _to_dev(inp)
    with torch.no_grad():
        lj = torch.nested.nested_tensor_from_jagged(d["lidar_values"], d["lidar_offsets"])
        enc = m.encoder(d["image"], lj)
        if precision != "bf16":                                      # bit-identity with the literal loop is a property of the launch chain
            m.decoder.fused_decode = False
        torch.cuda.synchronize(); t0 = time.time()
        toks, feats = m.generate(enc)
        torch.cuda.synchronize(); t_cached = time.time() - t0
        assert toks.shape == (2, 386) and feats.shape == (2, 385, 256)
        if precision != "bf16":                                      # the fused fp32 layer (r03, the default): same tokens, features to fp32 rounding
            m.decoder.fused_decode, m.decoder._decode_state = True, None
            torch.cuda.synchronize(); t0 = time.time()
            ftoks, ffeats = m.generate(enc)
            torch.cuda.synchronize(); t_fused = time.time() - t0
            print(f"\n[{precision}] fused decode layer: {t_fused:.2f} s against the launch chain's {t_cached:.2f} s")
            # FULL-length comparison: the fused layer sums its dot products in another order (features agree to ~2e-5), so a greedy
            # step whose two best logits lie closer than that may pick the other token - and only such a step may: the first divergence of every
            # sequence is located and the launch chain's own top-2 logit margin there must be a near-tie; up to it tokens are equal bit for bit
            for b_ in range(toks.shape[0]):
                diff = (ftoks[b_] != toks[b_]).nonzero()
                if diff.numel() == 0:
                    assert rel_err(ffeats[b_:b_ + 1].cpu(), feats[b_:b_ + 1].cpu()) < 1e-4
                    continue
                k = int(diff[0])                                     # token k was chosen from the logits of position k - 1
                lg, _ = m.decoder.predict(enc[b_:b_ + 1], toks[b_:b_ + 1, :k])
                top2 = lg[0].float().topk(2).values
                margin = float(top2[0] - top2[1])
                print(f"\n[{precision}] tile {b_}: fused decode leaves the launch chain's sequence at step {k} of 385, top-2 logit margin there {margin:.2e}")
                assert margin < 2e-4 * max(1.0, float(top2[0].abs())), (b_, k, margin)
                assert k >= 2 and torch.equal(ftoks[b_, :k], toks[b_, :k]) and rel_err(ffeats[b_:b_ + 1, :k - 1].cpu(), feats[b_:b_ + 1, :k - 1].cpu()) < 1e-4
        torch.cuda.synchronize(); t0 = time.time()
        ref_toks, ref_feats = m.generate(enc, steps=24, use_cache=False)
        torch.cuda.synchronize(); t_full24 = time.time() - t0
    print(f"\\n[{precision}] cached 385 steps: {t_cached:.2f} s; literal loop 24 steps: {t_full24:.2f} s (x{385 / 24:.0f} for 385)")
    assert torch.equal(toks[:, :25], ref_toks)
    if precision != "bf16":
        assert torch.equal(feats[:, :24], ref_feats[:, :24])
        ref = O.greedy_generate(O.encoder_fusion(inp["image"], inp["lidar_values"], inp["lidar_offsets"], sd), sd, steps=12)[0]
        assert torch.equal(toks[:, :13].cpu(), ref)


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_greedy_decode_of_the_early_fusion_model_equals_the_oracles_over_all_385_steps(precision):
    """One tile of the early-fusion model through all 385 greedy steps (launch chain, KV cache) against the oracle's literal loop on the host (VERDICT r05:
    the two-tile test above compares 12 steps).  Tokens must be equal up to the first step where the ORACLE's own top-2 logit margin falls below the
    mode's logit noise (a seeded random-init model produces near-ties; beyond such a step both sequences are valid and free-running) - and the test says
    how far that was."""
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd)
    inp = O.make_inputs(1, seed=11)
    d = _to_dev(inp)
    with torch.no_grad():
        enc = m.encoder(d["image"], (d["lidar_values"], d["lidar_offsets"]))
        m.decoder.fused_decode = False
        toks, _ = m.generate(enc)
        ref_enc = O.encoder_fusion(inp["image"], inp["lidar_values"], inp["lidar_offsets"], sd)
        # the oracle's loop, keeping its margins: Pix2PolyPredictor.test_generate (predictor_pix2poly.py:188-207)
        preds = torch.full((1, 1), O.BOS, dtype=torch.long)
        margins = []
        for _ in range(O.MAX_LEN - 1):
            logits, _f = O.decoder_predict(ref_enc, preds, sd)
            top2 = logits[0].float().topk(2).values
            margins.append(float(top2[0] - top2[1]) / max(1.0, float(top2[0].abs())))
            preds = torch.cat([preds, torch.softmax(logits, -1).argmax(-1, keepdim=True)], 1)
    noise = 1e-5 if precision == "fp32" else 2e-4                    # relative logit noise of the mode (measured: 1e-6 / 2e-5), with margin
    low = [k for k, mg in enumerate(margins) if mg < noise]
    upto = (low[0] + 1) if low else O.MAX_LEN                        # token k + 1 is chosen at step k
    print(f"\n[{precision}] oracle's first near-tie (margin < {noise:g}) at step {low[0] if low else None} of {O.MAX_LEN - 1}; tokens compared: {upto}")
    assert upto >= 13
    assert torch.equal(toks[:, :upto].cpu(), preds[:, :upto]), int((toks[0, :upto].cpu() != preds[0, :upto]).nonzero()[0])


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "bf16"])
def test_graph_replayed_decode_equals_eager_decode(precision):
    """generate(graphs=True): call 1 eager, call 2 captures one hipGraph per step, call 3+ only replays — tokens and features equal
    the eager KV-cache path on every call, also when the encoder features change between calls and after a weight update."""
    sd = O.make_state_dict("image", seed=42)
    m, cfg = _model("image", precision, sd)
    steps = 40
    with torch.no_grad():
        encs = [m.encoder(O.make_inputs(3, seed=s)["image"].to(DEV)) for s in (1, 2, 3, 4)]
        want = [m.generate(e, steps=steps) for e in encs]
        want = [(t.clone(), f.clone()) for t, f in want]
        for call, e in enumerate(encs):
            toks, feats = m.generate(e, steps=steps, graphs=True)
            assert torch.equal(toks, want[call][0]) and torch.equal(feats, want[call][1]), call
        assert m.decoder._decode_state["graphs"] is not None and len(m.decoder._decode_state["graphs"]) == steps
        b1 = m.decoder.decoder.layers[0].linear1.bias
        b1.data.add_(torch.randn_like(b1) * 0.5)                                            # weights changed: graphs must go
        ref = m.generate(encs[0], steps=steps)
        got = m.generate(encs[0], steps=steps, graphs=True)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        assert not torch.equal(ref[1], want[0][1])


def test_graph_replayed_decode_single_tile_and_step_limit():
    sd = O.make_state_dict("image", seed=42)
    m, cfg = _model("image", "bf16", sd)
    with torch.no_grad():
        enc = m.encoder(O.make_inputs(1, seed=2)["image"].to(DEV))
        ref = m.generate(enc, steps=6)
        for _ in range(3):
            got = m.generate(enc, steps=6, graphs=True)
            assert got[0].shape == (1, 7) and torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
        from pixelspointspolygons_amd._lib import P3Error
        with pytest.raises(P3Error):
            m.generate(enc, steps=386)


@pytest.mark.parametrize("precision", PARITY_MODES)
def test_lidar_feature_dropout_zeroes_the_whole_batch(precision):
    """SURVEY §9-8 (early_fusion_vit.py:113-119): one draw per BATCH; p = 1.0 (what validation sets) zeroes every tile's LiDAR
    features before the fusion conv, p = 0.0 never fires.  Checked against the oracle with the LiDAR map scaled by 0."""
    sd = O.make_state_dict("fusion", seed=42)
    inp = O.make_inputs(3, seed=21)
    d = _to_dev(inp)
    lidar = (d["lidar_values"], d["lidar_offsets"])
    with torch.no_grad():
        m1, _ = _model("fusion", precision, sd, lidar_dropout=1.0)
        got = m1.encoder(d["image"], lidar)
        ref = O.encoder_fusion(inp["image"], inp["lidar_values"], inp["lidar_offsets"], sd, lidar_scale=0.0)
        assert rel_err(got.float().cpu(), ref) < TOL32
        m0, _ = _model("fusion", precision, sd, lidar_dropout=0.0)
        mn, _ = _model("fusion", precision, sd)
        torch.manual_seed(0)
        a, b = m0.encoder(d["image"], lidar), mn.encoder(d["image"], lidar)
        assert torch.equal(a, b) and not torch.equal(a, got)


def test_lidar_only_model_accepts_all_three_lidar_input_forms():
    """nested jagged tensor (the reference's collate output), (values, offsets) pair (bench.py) and dense [B, N, 3]."""
    sd = O.make_state_dict("lidar", seed=42)
    m, cfg = _model("lidar", "fp32", sd)
    inp = O.make_inputs(2, seed=8, n_points=500, jitter=0)
    d = _to_dev(inp)
    y = d["y"][:, :-1]
    with torch.no_grad():
        a, _ = m(None, torch.nested.nested_tensor_from_jagged(d["lidar_values"], d["lidar_offsets"]), y)
        b, _ = m(None, (d["lidar_values"], d["lidar_offsets"]), y)
        c, _ = m(None, d["lidar_values"].view(2, 500, 3), y)
    assert torch.equal(a, b) and torch.equal(a, c)


@pytest.mark.parametrize("precision", ["fp32", "fp32x3", "bf16"])
def test_full_bench_batch_is_batch_independent_and_matches_the_oracle(precision):
    """BASELINE configs[2] at its FULL size (64 tiles, 3 k points each): in eval mode every tile's logits / permutation rows must not depend
    on its neighbours in the batch - tile b of the batch of 64 == the same tile run alone, bit for bit (row-independent GEMM / LayerNorm
    arithmetic, per-(tile, head) attention, per-tile pillar sort / Sinkhorn) - and two of the 64 tiles are checked against the oracle."""
    sd = O.make_state_dict("fusion", seed=42)
    m, cfg = _model("fusion", precision, sd, batch_size=64)
    inp = O.make_inputs(64, seed=1234)
    d = _to_dev(inp)
    off = inp["lidar_offsets"]
    with torch.no_grad():
        logits, perm = m(d["image"], (d["lidar_values"], d["lidar_offsets"]), d["y"][:, :-1])
        assert logits.shape == (64, 385, 227) and perm.shape == (64, 192, 192)
        assert torch.isfinite(logits).all() and torch.isfinite(perm).all()
        rows = perm.sum(-1)
        assert float((rows - 1).abs().max()) < 1e-3                       # softmax rows of the Sinkhorn output
        for b in (0, 17, 63):
            vals = d["lidar_values"][off[b]:off[b + 1]].contiguous()
            offs = torch.tensor([0, int(off[b + 1] - off[b])], device=DEV)
            l1, p1 = m(d["image"][b:b + 1], (vals, offs), d["y"][b:b + 1, :-1])
            assert torch.equal(l1[0], logits[b]) and torch.equal(p1[0], perm[b]), b
        tol = TOL32 if precision != "bf16" else TOL16
        for b in (17, 63):
            lo, hi = int(off[b]), int(off[b + 1])
            rl, rp = O.pix2poly_forward({k: v.clone() for k, v in sd.items()}, inp["y"][b:b + 1, :-1], inp["image"][b:b + 1],
                                        (inp["lidar_values"][lo:hi], torch.tensor([0, hi - lo])))
            assert rel_err(logits[b:b + 1].float().cpu(), rl) < tol and rel_err(perm[b:b + 1].float().cpu(), rp) < perm_tol(tol)
            if precision != "bf16":
                assert torch.equal(logits[b:b + 1].float().cpu().argmax(-1), rl.argmax(-1))


def test_two_models_of_different_precision_share_one_process():
    """The product precision belongs to the MODEL (P3_F32X3 per call, hip.scope_module / @precision_scoped), not to the process: an exact and an fp32x3 model
    built side by side, their forwards and backwards interleaved, each give bit for bit what they give alone."""
    from pixelspointspolygons_amd import hip
    from pixelspointspolygons_amd.training import pix2poly_loss
    sd = O.make_state_dict("image", seed=42)
    inp = O.make_inputs(2, seed=5)
    d = _to_dev(inp)
    y = d["y"]

    def alone(precision):
        m, _ = _model("image", precision, sd)
        m.train()
        m.decoder.set_dropout(0.0)
        logits, perm = m(d["image"], None, y[:, :-1])
        loss = pix2poly_loss(logits, perm, y[:, 1:], d["y_perm"])[0]
        loss.backward()
        return logits.detach().clone(), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}

    la, ga = alone("fp32")
    lb, gb = alone("fp32x3")
    assert not torch.equal(la, lb)                                  # the two modes are different arithmetic
    ma, _ = _model("image", "fp32", sd)
    mb, _ = _model("image", "fp32x3", sd)                           # built LAST: under the old process-global switch it would have decided for both
    for m in (ma, mb):
        m.train()
        m.decoder.set_dropout(0.0)
    assert ma.p3_split is False and mb.p3_split is True and not hip.split_now()
    l1, p1 = ma(d["image"], None, y[:, :-1])
    l2, p2 = mb(d["image"], None, y[:, :-1])
    assert not hip.split_now()
    loss2 = pix2poly_loss(l2, p2, y[:, 1:], d["y_perm"])[0]
    loss1 = pix2poly_loss(l1, p1, y[:, 1:], d["y_perm"])[0]
    loss1.backward()                                                # the exact model's backward runs between the fp32x3 model's forward and backward
    loss2.backward()
    assert torch.equal(l1.detach(), la) and torch.equal(l2.detach(), lb)
    atomic = {"bin_score", "decoder.embedding.weight"}              # summed with fp32 atomics in every mode: equal to rounding, not bit for bit
    for m, g in ((ma, ga), (mb, gb)):
        for k, p in m.named_parameters():
            if p.grad is not None:
                if k in atomic:
                    assert torch.allclose(p.grad, g[k], rtol=1e-5, atol=1e-6 * float(g[k].abs().max())), k
                else:
                    assert torch.equal(p.grad, g[k]), k
