"""Golden vectors for BASELINE configs[0] (scripts/predict_demo.py:8-50 -> predict/predictor_pix2poly.py:86-108,154-211): the demo
tile `demo_data/image0_CH_val.tif` of the reference repository through the image-only Pix2Poly model (ViT-S/8, seeded random weights -
no checkpoint is reachable offline), batch 1, 385-step greedy decode, Hungarian assignment, polygon assembly.

Run in the build container:  python tests/golden/make_demo_golden.py
  reads  /root/reference/demo_data/image0_CH_val.tif (PIL; 224 x 224 x 3 uint8 - the pixel bytes are stored in the fixture as DATA)
  writes tests/golden/demo_tile.npz: image_u8, tokens [1, 386], perm [1, 192, 192], flattened polygons, top-2 logit margins per step
The expected outputs come from the oracle (fp32 torch CPU restatement), whose Decoder / ScoreNet / predictor post-processing are pinned
against the reference's own classes by the other fixtures of this directory."""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import p3_oracle as O  # noqa: E402

TILE = "/root/reference/demo_data/image0_CH_val.tif"


def main():
    img_u8 = np.array(Image.open(TILE).convert("RGB"))
    assert img_u8.shape == (224, 224, 3) and img_u8.dtype == np.uint8
    # predictor.load_image_from_file: uint8 -> float32 / 255 -> normalize(mean 0, std 1) (predict/predictor.py:99-110)
    img = torch.from_numpy(img_u8).permute(2, 0, 1).unsqueeze(0).to(torch.float32) / 255.0
    sd = O.make_state_dict("image", O.VIT_S8, seed=42)
    with torch.no_grad():
        enc = O.encoder_vit(img, sd, cfg=O.VIT_S8)
        preds = torch.full((1, 1), O.BOS, dtype=torch.long)
        margins = []
        feats = None
        for _ in range(O.MAX_LEN - 1):
            logits, feats = O.decoder_predict(enc, preds, sd)
            top2 = torch.softmax(logits, -1).topk(2, dim=-1).values
            margins.append(float(top2[0, 0] - top2[0, 1]))
            preds = torch.cat([preds, torch.softmax(logits, -1).argmax(-1, keepdim=True)], 1)
        scores = O.scorenet(feats, sd, "scorenet1.") + O.scorenet(feats, sd, "scorenet2.").transpose(1, 2)
        perm = O.scores_to_permutations(scores)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "demo_tile.npz"), image_u8=img_u8, tokens=preds.numpy(), perm=perm.numpy().astype(np.uint8),
                        scores=scores.numpy(), margins=np.array(margins, dtype=np.float32), enc_sample=enc[0, ::97, ::31].numpy())
    print("tokens[:20]", preds[0, :20].tolist(), "EOS at", (preds[0] == O.EOS).nonzero().view(-1).tolist()[:3], "min top-2 margin", min(margins))


if __name__ == "__main__":
    main()
