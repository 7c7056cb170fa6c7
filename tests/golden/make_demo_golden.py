"""Golden vectors for BASELINE configs[0] (scripts/predict_demo.py:8-50 -> predict/predictor_pix2poly.py:86-108,154-211): the demo
tile `demo_data/image0_CH_val.tif` of the reference repository through the image-only Pix2Poly model (ViT-S/8), batch 1, 385-step
greedy decode, Hungarian assignment, polygon assembly.

No checkpoint is reachable offline and a random-init decoder emits one token for ever, so the fixture PLANTS a small trained part:
the seeded random model (oracle.make_state_dict("image", seed 42)) with three compact tensors - decoder.output.{weight, bias} and
decoder.decoder_pos_embed (156 k values) - fitted on the CPU (teacher forcing, Adam) to a sequence of six building outlines + EOS for
this tile.  That gives sharp, trained-like logits (argmax margins are stored), a real EOS and non-trivial polygons.  Only those three
tensors travel (fp32, in the fixture); everything else is regenerated from the seed.

Run in the build container:  python tests/golden/make_demo_golden.py
  reads  /root/reference/demo_data/image0_CH_val.tif (PIL; 224 x 224 x 3 uint8 - the pixel bytes are stored in the fixture as DATA)
  writes tests/golden/demo_tile.npz: image_u8, planted tensors, tokens [1, 386], perm [1, 192, 192], polygons, top-2 margins per step
The expected outputs come from the oracle (fp32 torch CPU restatement), whose Decoder / ScoreNet / predictor post-processing are pinned
against the reference's own classes by the other fixtures of this directory."""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import p3_oracle as O  # noqa: E402

TILE = "/root/reference/demo_data/image0_CH_val.tif"
PLANTED = ("decoder.output.weight", "decoder.output.bias", "decoder.decoder_pos_embed")
# six axis-aligned-ish building outlines (y, x pixel corners), the kind of target the dataset holds for such a tile
BUILDINGS = [[(20, 24), (20, 71), (58, 71), (58, 24)], [(30, 120), (30, 190), (66, 190), (66, 150), (52, 150), (52, 120)],
             [(96, 18), (96, 60), (140, 60), (140, 18)], [(110, 100), (110, 160), (150, 160), (150, 100)],
             [(170, 30), (170, 96), (204, 96), (204, 30)], [(168, 140), (168, 206), (210, 206), (210, 176), (190, 176), (190, 140)]]


def target_tokens():
    toks = [O.BOS]
    for poly in BUILDINGS:
        for y, x in poly:
            toks += [int(round(y / 224 * (O.NUM_BINS - 1))), int(round(x / 224 * (O.NUM_BINS - 1)))]
    return toks + [O.EOS]


def main():
    torch.manual_seed(0)
    img_u8 = np.array(Image.open(TILE).convert("RGB"))
    assert img_u8.shape == (224, 224, 3) and img_u8.dtype == np.uint8
    # predictor.load_image_from_file: uint8 -> float32 / 255 -> normalize(mean 0, std 1) (predict/predictor.py:99-110)
    img = torch.from_numpy(img_u8).permute(2, 0, 1).unsqueeze(0).to(torch.float32) / 255.0
    sd = O.make_state_dict("image", O.VIT_S8, seed=42)
    with torch.no_grad():
        enc = O.encoder_vit(img, sd, cfg=O.VIT_S8)
    # ---- plant: fit the three compact tensors by teacher forcing (inputs padded exactly like Decoder.predict pads them)
    tgt = torch.tensor([target_tokens()])
    L = tgt.shape[1]
    inp = torch.cat([tgt[:, :-1], torch.full((1, O.MAX_LEN - 1 - (L - 1)), O.PAD, dtype=torch.long)], 1)
    for k in PLANTED:
        sd[k] = sd[k].clone().requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in PLANTED], lr=3e-3)
    for it in range(400):
        logits, _ = O.decoder_forward(enc, inp, sd)
        lg = logits[0, :L - 1]
        loss = F.cross_entropy(lg, tgt[0, 1:]) + 0.5 * F.relu(1.0 - (lg.gather(1, tgt[0, 1:, None]) - lg.scatter(1, tgt[0, 1:, None], -1e9).max(1, keepdim=True).values)).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        ok = bool((lg.argmax(-1) == tgt[0, 1:]).all())
        if it % 25 == 0 or ok:
            print(it, float(loss), int((lg.argmax(-1) == tgt[0, 1:]).sum()), "/", L - 1, flush=True)
        if ok and float(loss) < 0.05:
            break
    planted = {k: sd[k].detach().clone() for k in PLANTED}
    for k in PLANTED:
        sd[k] = planted[k]
    # ---- the predictor's loop, literally (full re-run per step), recording the top-2 probability margin of every argmax
    with torch.no_grad():
        preds = torch.full((1, 1), O.BOS, dtype=torch.long)
        margins, feats = [], None
        for _ in range(O.MAX_LEN - 1):
            logits, feats = O.decoder_predict(enc, preds, sd)
            top2 = torch.softmax(logits, -1).topk(2, dim=-1).values
            margins.append(float(top2[0, 0] - top2[0, 1]))
            preds = torch.cat([preds, torch.softmax(logits, -1).argmax(-1, keepdim=True)], 1)
        scores = O.scorenet(feats, sd, "scorenet1.") + O.scorenet(feats, sd, "scorenet2.").transpose(1, 2)
        perm = O.scores_to_permutations(scores)
    assert preds[0, :L].tolist() == tgt[0].tolist(), "greedy decode does not reproduce the planted sequence"
    # polygons through the product-independent restatement of the predictor's post-processing
    dec = lambda t: (np.asarray(t[1:-1]).reshape(-1, 2).astype("float32") / (O.NUM_BINS - 1)) * 224.0       # Tokenizer.decode
    coords = O.predictor_postprocess(preds, lambda t: dec(t[t != O.PAD].numpy()))[0]
    idx, chains = O.permutation_polygons(perm[0])
    padded = np.full((O.MAX_VERTS, 2), float(O.PAD), dtype=np.float32)
    padded[: len(coords)] = coords
    polys = []
    for ch in chains:
        p = padded[[idx[i] for i in ch]][:, ::-1]
        p = p[p[:, 0] != float(O.PAD)]
        if len(p):
            polys.append(p)
    flat = np.concatenate(polys) if polys else np.zeros((0, 2), np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "demo_tile.npz"), image_u8=img_u8, tokens=preds.numpy(), perm=perm.numpy().astype(np.uint8),
                        scores=scores.numpy(), margins=np.array(margins, dtype=np.float32), poly_flat=flat, poly_len=np.array([len(p) for p in polys]),
                        target=tgt.numpy(), **{"planted." + k: v.numpy() for k, v in planted.items()})
    print("tokens[:30]", preds[0, :30].tolist(), "EOS at", (preds[0] == O.EOS).nonzero().view(-1).tolist()[:3], "min top-2 margin", min(margins),
          "polygons", [len(p) for p in polys])


if __name__ == "__main__":
    main()
