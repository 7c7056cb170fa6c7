"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules.

Runs only in the build container (needs /root/reference).  Usage:  python tests/golden/make_golden.py
Fixtures are data only: seeded inputs, seeded weights (reduced-shape cases) and the reference's outputs.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from transformers import ViTConfig, ViTModel  # noqa: E402  (before the timm stub is installed)
from _ref_import import load_reference  # noqa: E402
from oracle import p3_oracle as O  # noqa: E402

torch.set_grad_enabled(False)
ref = load_reference()


def npz(name, **kw):
    out = {}
    for k, v in kw.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, len(out), "arrays")


def sub(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def small_cfg():
    return dict(dim=64, depth=2, heads=2, mlp=128, patch=8, img=32, eps=1e-6)


def tokens(B, n_vertices, g, first_pad=None):
    L = 2 * n_vertices + 1
    y = torch.full((B, L), O.PAD, dtype=torch.long)
    for b in range(B):
        n = int(torch.randint(2, n_vertices, (1,), generator=g))
        y[b, 0] = O.BOS
        y[b, 1:1 + 2 * n] = torch.randint(0, O.NUM_BINS, (2 * n,), generator=g)
        y[b, 1 + 2 * n] = O.EOS
    return y


# ---------------------------------------------------------------- 1. Decoder (reduced + full)
def gen_decoder(tag, dim, heads, layers, nv, enc_len, B, ffn_note=2048, seed=7):
    g = torch.Generator().manual_seed(seed)
    sd = O.make_state_dict("image", dict(small_cfg(), img=int(enc_len ** 0.5) * 8), seed=seed, n_vertices=nv,
                           dec_dim=dim, dec_layers=layers)
    dec = ref.Decoder(vocab_size=O.VOCAB, encoder_len=enc_len, dim=dim, num_heads=heads, num_layers=layers,
                      max_len=2 * nv + 2, pad_idx=O.PAD)
    dec.load_state_dict(sub(sd, "decoder."), strict=True)
    dec.eval()
    enc = torch.randn(B, enc_len, dim, generator=g)
    y = tokens(B, nv, g)
    logits, feats = dec(enc, y)
    pl, pf = dec.predict(enc, y[:, :5])
    keep = {k: v for k, v in sd.items() if k.startswith("decoder.")} if dim <= 64 else {}
    npz(f"decoder_{tag}.npz", enc=enc, y=y, logits=logits, feats=feats, pred_logits=pl, pred_feats=pf,
        meta=np.array([dim, heads, layers, nv, enc_len, seed]), **{"w::" + k: v for k, v in keep.items()})
    return sd, dec


sd_s, dec_s = gen_decoder("small", 64, 8, 2, 10, 16, 3)
sd_f, dec_f = gen_decoder("full", 256, 8, 6, 192, 784, 1, seed=42)

# ---------------------------------------------------------------- 2. greedy decode (reduced): bit-exact tokens
g = torch.Generator().manual_seed(11)
enc = torch.randn(2, 16, 64, generator=g)
preds = torch.full((2, 1), O.BOS, dtype=torch.long)
for i in range(2 * 10 + 1):
    lg, feats = dec_s.predict(enc, preds)
    nxt = torch.softmax(lg, -1).argmax(-1, keepdim=True)     # predictor_pix2poly.py:165,196-197
    preds = torch.cat([preds, nxt], 1)
npz("greedy_small.npz", enc=enc, tokens=preds, feats=feats)


# ---------------------------------------------------------------- 2b. greedy decode at head_dim 32 (dim 256, 2 layers): bit-exact tokens
sd_g = O.make_state_dict("image", dict(small_cfg(), img=32), seed=77, n_vertices=10, dec_dim=256, dec_layers=2)
dec_g = ref.Decoder(vocab_size=O.VOCAB, encoder_len=16, dim=256, num_heads=8, num_layers=2, max_len=22, pad_idx=O.PAD)
dec_g.load_state_dict(sub(sd_g, "decoder."), strict=True)
dec_g.eval()
g = torch.Generator().manual_seed(78)
enc = torch.randn(3, 16, 256, generator=g)
preds = torch.full((3, 1), O.BOS, dtype=torch.long)
for i in range(21):
    lg, feats = dec_g.predict(enc, preds)
    preds = torch.cat([preds, torch.softmax(lg, -1).argmax(-1, keepdim=True)], 1)
yg = tokens(3, 10, g)
lgf, ftf = dec_g(enc, yg)
npz("greedy_d256.npz", enc=enc, tokens=preds, feats=feats, y=yg, logits=lgf, fwd_feats=ftf,
    wsum=np.array([float(sum(v.double().sum() for k, v in sd_g.items() if k.startswith("decoder.")))]))

# ---------------------------------------------------------------- 3. ScoreNet (reduced dims + full, eval and train BN)
for tag, dim, nv, B, seed in (("small", 64, 10, 3, 7), ("full", 256, 192, 1, 42)):
    sd = sd_s if tag == "small" else sd_f
    g = torch.Generator().manual_seed(seed + 1)
    feats = torch.randn(B, 2 * nv + 1, dim, generator=g)
    outs = {}
    for mode in ("eval", "train"):
        for s in ("scorenet1.", "scorenet2."):
            net = ref.ScoreNet(nv, in_channels=2 * dim)
            net.load_state_dict(sub(sd, s), strict=True)
            net.train(mode == "train")
            outs[f"{s}{mode}"] = net(feats)
            if mode == "train":
                outs[f"{s}rm1"] = net.bn1.running_mean
                outs[f"{s}rv1"] = net.bn1.running_var
    keep = {k: v for k, v in sd.items() if k.startswith("scorenet")} if tag == "small" else {}
    npz(f"scorenet_{tag}.npz", feats=feats, meta=np.array([dim, nv, seed]), **outs,
        **{"w::" + k: v for k, v in keep.items()})

# ---------------------------------------------------------------- 4. Sinkhorn
g = torch.Generator().manual_seed(5)
sc_s = torch.randn(2, 12, 12, generator=g) * 3
sc_f = torch.randn(1, 192, 192, generator=g) * 2
alpha = torch.tensor(1.0)
npz("sinkhorn.npz", scores_small=sc_s, lot_small=ref.log_optimal_transport(sc_s, alpha, 100),
    scores_full=sc_f, lot_full=ref.log_optimal_transport(sc_f, alpha, 100),
    lot_small_it3=ref.log_optimal_transport(sc_s, torch.tensor(0.3), 3))

# ---------------------------------------------------------------- 5. EncoderDecoder.forward + EarlyFusionViT.forward glue (hybrid)
class _NS(types.SimpleNamespace):
    pass


def ns(d):
    return _NS(**{k: ns(v) if isinstance(v, dict) else v for k, v in d.items()})


cfgd = dict(experiment=dict(encoder=dict(use_images=True, use_lidar=True, out_feature_dim=64),
                            model=dict(tokenizer=dict(max_num_vertices=10), sinkhorn_iterations=100),
                            lidar_dropout=None))
cfg = ns(cfgd)
vc = small_cfg()
sd = O.make_state_dict("fusion", vc, seed=3, n_vertices=10, dec_dim=64, dec_layers=2)


class _ImgEmbed(torch.nn.Module):
    def forward(self, x):
        return O.patch_embed(x, sd, "encoder.image_embed.", vc["patch"])


class _LidarEmbed(torch.nn.Module):
    def forward(self, x, return_flattened=True):
        out = O.pillar_stem(x.values(), x.offsets(), sd, "encoder.lidar_embed.", grid=(4, 4), voxel=(8.0, 8.0, 100.0),
                            max_points=8, max_voxels=16, training=False)
        return out.flatten(2).transpose(1, 2) if return_flattened else out


class _Vit(torch.nn.Module):
    def forward(self, x):
        return O.vit_blocks(x, sd, "encoder.vit.", vc["depth"], vc["heads"], vc["eps"])


class _Fusion(torch.nn.Module):
    def forward(self, x):
        import torch.nn.functional as F
        x = F.conv2d(x, sd["encoder.fusion_layer.0.weight"], sd["encoder.fusion_layer.0.bias"], padding=1)
        return F.relu(O._bn(x, sd, "encoder.fusion_layer.1", False, 1e-5, 0.1, dims=(0, 2, 3)))


EF = ref.EarlyFusionViT
enc_mod = EF.__new__(EF)
torch.nn.Module.__init__(enc_mod)
enc_mod.cfg = cfg
enc_mod.image_embed, enc_mod.lidar_embed, enc_mod.vit, enc_mod.fusion_layer = _ImgEmbed(), _LidarEmbed(), _Vit(), _Fusion()
enc_mod.bottleneck = torch.nn.AdaptiveAvgPool1d(64)
dec = ref.Decoder(vocab_size=O.VOCAB, encoder_len=16, dim=64, num_heads=8, num_layers=2, max_len=22, pad_idx=O.PAD)
dec.load_state_dict(sub(sd, "decoder."))
model = ref.EncoderDecoder(enc_mod, dec, cfg)
for s in ("scorenet1", "scorenet2"):
    net = ref.ScoreNet(10, in_channels=128)
    net.load_state_dict(sub(sd, s + "."))
    setattr(model, s, net)
model.bin_score.data.fill_(1.0)
model.eval()
inp = O.make_inputs(2, seed=99, n_points=60, jitter=10, n_vertices=10, img_size=32, min_verts=3)
lidar = torch.nested.nested_tensor_from_jagged(inp["lidar_values"], inp["lidar_offsets"])
seq, perm = model(inp["image"], lidar, inp["y"][:, :-1])
npz("encdec_small.npz", image=inp["image"], lidar_values=inp["lidar_values"], lidar_offsets=inp["lidar_offsets"],
    y=inp["y"], seq_pred=seq, perm_mat=perm, **{"w::" + k: v for k, v in sd.items()})

# ---------------------------------------------------------------- 6. ViT body vs the independent transformers.ViTModel
def hf_vit(vc, sd, img):
    c = ViTConfig(hidden_size=vc["dim"], num_hidden_layers=vc["depth"], num_attention_heads=vc["heads"],
                  intermediate_size=vc["mlp"], patch_size=vc["patch"], image_size=vc["img"], layer_norm_eps=vc["eps"],
                  hidden_act="gelu", hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, qkv_bias=True)
    m = ViTModel(c, add_pooling_layer=False).eval()
    D = vc["dim"]
    t = {"embeddings.cls_token": sd["encoder.vit.cls_token"], "embeddings.position_embeddings": sd["encoder.vit.pos_embed"],
         "embeddings.patch_embeddings.projection.weight": sd["encoder.vit.patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["encoder.vit.patch_embed.proj.bias"],
         "layernorm.weight": sd["encoder.vit.norm.weight"], "layernorm.bias": sd["encoder.vit.norm.bias"]}
    for i in range(vc["depth"]):
        p, q = f"encoder.vit.blocks.{i}.", f"layers.{i}."
        W, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nme in enumerate(("q_proj", "k_proj", "v_proj")):
            t[q + f"attention.{nme}.weight"] = W[j * D:(j + 1) * D]
            t[q + f"attention.{nme}.bias"] = b[j * D:(j + 1) * D]
        t[q + "attention.o_proj.weight"] = sd[p + "attn.proj.weight"]
        t[q + "attention.o_proj.bias"] = sd[p + "attn.proj.bias"]
        t[q + "layernorm_before.weight"] = sd[p + "norm1.weight"]
        t[q + "layernorm_before.bias"] = sd[p + "norm1.bias"]
        t[q + "layernorm_after.weight"] = sd[p + "norm2.weight"]
        t[q + "layernorm_after.bias"] = sd[p + "norm2.bias"]
        t[q + "mlp.fc1.weight"] = sd[p + "mlp.fc1.weight"]
        t[q + "mlp.fc1.bias"] = sd[p + "mlp.fc1.bias"]
        t[q + "mlp.fc2.weight"] = sd[p + "mlp.fc2.weight"]
        t[q + "mlp.fc2.bias"] = sd[p + "mlp.fc2.bias"]
    missing = m.load_state_dict(t, strict=False)
    assert not missing.unexpected_keys, missing
    assert not missing.missing_keys, missing
    return m(pixel_values=img).last_hidden_state


vcs = small_cfg()
sdv = O.make_state_dict("image", vcs, seed=21, n_vertices=10, dec_dim=64, dec_layers=1)
g = torch.Generator().manual_seed(22)
img = torch.rand(2, 3, 32, 32, generator=g)
npz("vit_hf_small.npz", image=img, tokens=hf_vit(vcs, sdv, img),
    **{"w::" + k: v for k, v in sdv.items() if k.startswith("encoder.")})
sdF = O.make_state_dict("image", O.VIT_S8, seed=42)
imgF = torch.rand(1, 3, 224, 224, generator=g).to(torch.float16).float()  # stored as f16 (exactly representable)
outF = hf_vit(O.VIT_S8, sdF, imgF)
npz("vit_hf_s8.npz", image_seed=np.array([22]), image=imgF.to(torch.float16), tokens=outF[:, ::8, :],
    wsum=np.array([float(sum(v.double().sum() for k, v in sdF.items() if v.is_floating_point()))]))

# ---------------------------------------------------------------- 7. Tokenizer
import importlib  # noqa: E402
tokmod = importlib.import_module("pixelspointspolygons.models.pix2poly.tokenizer")
tcfg = ns(dict(experiment=dict(model=dict(tokenizer=dict(num_bins=224, max_num_vertices=192, pad_idx=None, max_len=None,
                                                          generation_steps=None)),
                               encoder=dict(in_width=224, in_height=224)), run_type=dict(name="debug")))
tk = tokmod.Tokenizer(tcfg)
coords = np.random.RandomState(0).rand(7, 2) * 224
toks, idx = tk(coords.copy(), shuffle=False)
dec_c = tk.decode(torch.tensor(toks))
npz("tokenizer.npz", coords=coords, tokens=np.array(toks), decoded=dec_c,
    consts=np.array([tk.BOS_code, tk.EOS_code, tk.PAD_code, tk.vocab_size, tk.max_len,
                     tcfg.experiment.model.tokenizer.generation_steps]))
print("done")
