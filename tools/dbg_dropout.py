import sys, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from oracle import p3_oracle as O
from helpers import rel_err
from pixelspointspolygons_amd import hip, ops
from pixelspointspolygons_amd.config import make_config
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
DEV = "cuda"
sd = O.make_state_dict("image", seed=42)
inp = O.make_inputs(2, seed=55)
probs = lambda site: 0.05 if site >= 250 else 0.1
SEED = 20260101
seed = torch.full((1,), SEED, dtype=torch.int64, device=DEV)
def masks(site, shape):
    return hip.dropout_apply(torch.ones(shape, dtype=torch.float32, device=DEV), torch.float32, (seed, site, probs(site))).cpu().double()
cfg = make_config("vit", precision="fp32", device=DEV)
m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
m.load_state_dict(sd, strict=True)
m.train()
ops.manual_seed(SEED, DEV)
d = {k: v.to(DEV) for k, v in inp.items()}
with torch.no_grad():
    enc = m.encoder(d["image"])
    encr = O.encoder_vit(inp["image"], sd)
    print("enc", rel_err(enc.float().cpu(), encr))
    for only in [None, 250, 251, 0, 1, 2, 3, 4, 5]:
        mk = (lambda site, shape: masks(site, shape) if (only is None or site == only) else torch.ones(shape, dtype=torch.double))
        # product with only that site active
        for li, lyr in enumerate(m.decoder.decoder.layers):
            pass
        m.decoder.set_dropout(0.0)
        def setp(site, p):
            dec = m.decoder
            if site == 250: dec.decoder_pos_drop.p = p
            elif site == 251: dec.encoder_pos_drop.p = p
            else:
                lyr = dec.decoder.layers[site // 8]
                k = site % 8
                if k == 0: lyr.self_attn.dropout = p
                elif k == 1: lyr.dropout1.p = p
                elif k == 2: lyr.multihead_attn.dropout = p
                elif k == 3: lyr.dropout2.p = p
                elif k == 4: lyr.dropout.p = p
                elif k == 5: lyr.dropout3.p = p
        if only is None:
            for s in [250, 251] + [8 * i + k for i in range(6) for k in range(6)]:
                setp(s, probs(s))
        else:
            setp(only, probs(only))
        logits, feats = m.decoder(enc, d["y"][:, :-1])
        lr, fr = O.decoder_forward(encr.double(), inp["y"][:, :-1], {k: v.double() if v.is_floating_point() else v for k, v in sd.items()}, masks=mk)
        print("site", only, "logits", rel_err(logits.float().cpu(), lr), "feats", rel_err(feats.float().cpu(), fr))
