"""Golden vectors for the Pix2Poly predictor's host post-processing (predict/predictor_pix2poly.py: postprocess :284-305,
permutations_to_polygons :213-282, coord_and_perm_to_polygons :111-138) produced by the REFERENCE's own methods.  Build container only;
absent third-party packages are auto-stubbed as in make_ffl_loss_golden.py (none of them is on this code path).
Usage: python tests/golden/make_postprocess_golden.py"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_ffl_loss_golden import _Finder  # noqa: E402

REF = "/root/reference"


def load_predictor():
    f = _Finder()
    f.ROOTS = f.ROOTS + ("hydra", "omegaconf", "timm", "open3d", "laspy", "copclib", "torchvision", "transformers", "wandb", "tqdm", "PIL", "colorlog",
                         "albumentations", "pandas", "sklearn", "scipy_disabled", "huggingface_hub", "geopandas", "affine")
    sys.meta_path.append(f)
    base = REF + "/pixelspointspolygons"
    for name, path in (("pixelspointspolygons", base), ("pixelspointspolygons.predict", base + "/predict"), ("pixelspointspolygons.models", base + "/models"),
                       ("pixelspointspolygons.models.pix2poly", base + "/models/pix2poly"), ("pixelspointspolygons.misc", base + "/misc"),
                       ("pixelspointspolygons.datasets", base + "/datasets"), ("pixelspointspolygons.eval", base + "/eval")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    for stub in ("pixelspointspolygons.misc", "pixelspointspolygons.datasets", "pixelspointspolygons.eval", "pixelspointspolygons.models.pix2poly"):
        m = sys.modules[stub]
        m.__getattr__ = lambda name: type(name, (object,), {})          # whatever the predictor imports from its siblings
    sys.modules["pixelspointspolygons.predict.predictor"] = types.ModuleType("pixelspointspolygons.predict.predictor")
    sys.modules["pixelspointspolygons.predict.predictor"].Predictor = object
    import importlib.util
    spec = importlib.util.spec_from_file_location("pixelspointspolygons.predict.predictor_pix2poly", base + "/predict/predictor_pix2poly.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    spec = importlib.util.spec_from_file_location("ref_tokenizer", base + "/models/pix2poly/tokenizer.py")
    tk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tk)
    return mod, tk


def ns(**kw):
    return types.SimpleNamespace(**kw)


def main():
    mod, tk = load_predictor()
    P = [getattr(mod, n) for n in dir(mod) if isinstance(getattr(mod, n), type) and hasattr(getattr(mod, n), "permutations_to_polygons")][0]
    cfg = ns(experiment=ns(model=ns(tokenizer=ns(max_num_vertices=192, num_bins=224)), encoder=ns(in_width=224, in_height=224)))
    tokenizer = tk.Tokenizer(cfg)                      # sets cfg...tokenizer.pad_idx / max_len / generation_steps
    self = ns(tokenizer=tokenizer, cfg=cfg)
    self.postprocess = lambda bp: P.postprocess(self, bp)
    self.permutations_to_polygons = lambda perm, graph, out="torch": P.permutations_to_polygons(self, perm, graph, out=out)
    g = torch.Generator().manual_seed(7)
    B, N, L = 5, 192, 386
    preds = torch.full((B, L), tokenizer.PAD_code, dtype=torch.long)
    perm = torch.zeros(B, N, N)
    nverts = [13, 0, 40, 7, 192]
    for b, n in enumerate(nverts):
        preds[b, 0] = tokenizer.BOS_code
        preds[b, 1:1 + 2 * n] = torch.randint(0, 224, (2 * n,), generator=g)
        preds[b, 1 + 2 * n] = tokenizer.EOS_code
        i = 0
        while i < n:                                   # closed polygons over the first n vertices (3..9 vertices each), identity elsewhere
            ln = min(int(torch.randint(3, 10, (1,), generator=g)), n - i)
            order = i + torch.randperm(ln, generator=g)
            for k in range(ln):
                perm[b, order[k], order[(k + 1) % ln]] = 1.0
            i += ln
        for k in range(n, N):
            perm[b, k, k] = 1.0
    preds[3, 1 + 2 * 7] = 5                           # EOS at an odd offset afterwards -> sanity check rejects the sample
    preds[3, 2 + 2 * 7] = tokenizer.EOS_code
    coords = P.postprocess(self, preds)
    polys = P.coord_and_perm_to_polygons(self, preds, perm)
    arrays = {"preds": preds.numpy(), "perm": perm.numpy(), "n_tiles": np.int64(B)}
    for b in range(B):
        arrays[f"coords{b}"] = np.zeros((0, 2)) if coords[b] is None else np.asarray(coords[b])
        arrays[f"coords{b}_none"] = np.int64(coords[b] is None)
        arrays[f"npoly{b}"] = np.int64(len(polys[b]))
        for k, p in enumerate(polys[b]):
            arrays[f"poly{b}_{k}"] = p.numpy()
    graph = [torch.rand(N, 2, generator=g) * 224 for _ in range(B)]
    for fmt in ("numpy", "list", "coco"):
        res = P.permutations_to_polygons(self, perm, [x.clone() for x in graph], out=fmt)
        for b in range(B):
            arrays[f"{fmt}_n{b}"] = np.int64(len(res[b]))
            for k, p in enumerate(res[b]):
                arrays[f"{fmt}{b}_{k}"] = np.asarray(p, dtype=np.float64)
    arrays["graph"] = torch.stack(graph).numpy()
    np.savez_compressed(os.path.join(HERE, "postprocess.npz"), **arrays)
    print("wrote postprocess.npz", len(arrays), "arrays; polygons per tile:", [len(p) for p in polys])


if __name__ == "__main__":
    main()
