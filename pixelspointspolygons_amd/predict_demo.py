"""BASELINE configs[0] on the HIP path: `scripts/predict_demo.py +image_file=demo_data/image0_CH_val.tif` with the image-only Pix2Poly
model (reference: scripts/predict_demo.py:8-50 -> Predictor.predict_file, predict/predictor_pix2poly.py:86-108 ->
load_image_from_file, predict/predictor.py:99-110 -> batch_to_polygons, predictor_pix2poly.py:141-211).

  image_tensor(tile_u8, cfg, device)      uint8 HWC tile -> fp32 [1, 3, H, W] exactly as load_image_from_file builds it
  predict_tile(model, tokenizer, tile)    -> (polygons, tokens): encoder, KV-cached 385-step greedy decode (hipGraph replay), both
                                             ScoreNets, device Hungarian assignment, polygon assembly
  demo_tile() / demo_model(...)           the fixture tile (pixel bytes of the reference's demo_data tile, tests/golden/demo_tile.npz) and
                                          the seeded model with the fixture's planted output layer, used by tests/ and by bench.py's
                                          `predict` leg (no checkpoint is reachable offline)
File decoding (rasterio) and plotting stay the reference's predictor code."""
import os
import time

import numpy as np
import torch

from . import postprocess
from .config import make_config

_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "demo_tile.npz")


def image_tensor(tile_u8, cfg, device):
    """predictor.py:103-110: HWC uint8 -> [1, 3, H, W] float32 / 255, then torchvision normalize with the encoder's mean / std."""
    enc = cfg.experiment.encoder
    x = torch.from_numpy(np.ascontiguousarray(tile_u8)).permute(2, 0, 1).unsqueeze(0).to(device).to(torch.float32) / 255.0
    mean = torch.tensor(list(enc.image_mean), dtype=torch.float32, device=x.device).view(1, -1, 1, 1)
    std = torch.tensor(list(enc.image_std), dtype=torch.float32, device=x.device).view(1, -1, 1, 1)
    return (x - mean) / std


@torch.no_grad()
def predict_tile(model, tokenizer, tile_u8, graphs=True):
    """-> (list of [n, 2] (x, y) polygons in pixels, token tensor [1, 386] on the host)"""
    x = image_tensor(tile_u8, model.cfg, model.cfg.host.device)
    feats = model.encoder(x)
    tokens, dec_feats = model.generate(feats, graphs=graphs)
    perm = model.permutations(dec_feats)
    polys = postprocess.coord_and_perm_to_polygons(tokens.cpu(), perm.cpu(), tokenizer, model.max_num_vertices)[0]
    return polys, tokens.cpu()


def demo_tile():
    """-> (tile uint8 [224, 224, 3], fixture dict or None).  Without the fixture (a stripped checkout) a synthetic tile is returned."""
    if os.path.exists(_FIXTURE):
        d = np.load(_FIXTURE)
        return d["image_u8"], d
    g = np.random.default_rng(0)
    return g.integers(0, 256, size=(224, 224, 3), dtype=np.uint8), None


def demo_model(device, precision="fp32", state_dict=None, fixture=None):
    """image-only Pix2Poly (config/experiment/p2p_image.yaml shape: ViT-S/8) + its tokenizer; `state_dict` (reference-keyed) is loaded
    when given, then the fixture's planted tensors are written over it."""
    from .pix2poly import Pix2PolyModel, Tokenizer
    cfg = make_config("vit", precision=precision, device=device, batch_size=1)
    tk = Tokenizer(cfg)
    torch.manual_seed(42)
    model = Pix2PolyModel(cfg, tk.vocab_size, 0)
    if state_dict is not None:
        model.load_state_dict(state_dict, strict=True)
    if fixture is not None:
        sd = model.state_dict()
        with torch.no_grad():
            for k in fixture.files:
                if k.startswith("planted."):
                    sd[k[len("planted."):]].copy_(torch.from_numpy(fixture[k]))
    model.eval()
    return model, tk


def predict_leg(device, repeats=5):
    """bench.py `predict` object: s/tile of the whole predict path at batch 1 (graphs on: first call eager, second captures the 385
    per-step graphs, later calls replay), in the fp32 parity mode and in bf16.  The model is the one tests/test_predict_demo_gpu.py checks
    against the oracle: seed-42 weights (synthetic.make_state_dict) + the fixture's planted, CPU-fitted output layer, so the decode ends in a
    real EOS and the timed tail (2 x ScoreNet, Hungarian assignment, polygon assembly) works on real polygons."""
    from . import synthetic
    tile, fx = demo_tile()
    out = {"tile": "demo_data/image0_CH_val.tif (pixel bytes from tests/golden/demo_tile.npz)" if fx is not None else "synthetic (fixture missing)",
           "batch": 1, "decode_steps": 385, "what": "image prep + ViT-S/8 encoder + KV-cached greedy decode (one hipGraph per step) + 2x ScoreNet + "
           "device Hungarian assignment + polygon assembly; seed-42 weights with the demo fixture's planted output layer (trained-like logits, "
           "a real EOS, non-trivial polygons; token / polygon parity of exactly this model: tests/test_predict_demo_gpu.py)"}
    sd = synthetic.make_state_dict("image", seed=42) if fx is not None else None
    want = torch.from_numpy(fx["tokens"]) if fx is not None else None
    for prec in ("fp32", "bf16"):
        model, tk = demo_model(device, prec, state_dict=sd, fixture=fx)
        for _ in range(3):
            polys, tokens = predict_tile(model, tk, tile)
        torch.cuda.synchronize()
        ts = []
        for _ in range(repeats):
            t0 = time.perf_counter()
            polys, tokens = predict_tile(model, tk, tile)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        has_eos = bool((tokens[0] == tk.EOS_code).any())
        eos = int((tokens[0] == tk.EOS_code).nonzero()[0]) if has_eos else None
        out[prec] = {"s_per_tile": round(float(np.median(ts)), 4), "min_s_per_tile": round(min(ts), 4), "polygons": len(polys),
                     "vertices": int(sum(len(p) for p in polys)), "eos_at": eos}
        if want is not None and eos is not None:      # tokens up to the planted EOS against the fixture (the oracle's sequence)
            n = int((want[0] == tk.EOS_code).nonzero()[0]) + 1
            out[prec]["tokens_equal_fixture_upto_eos"] = bool(torch.equal(tokens[0, :n], want[0, :n]))
        del model
    return out
