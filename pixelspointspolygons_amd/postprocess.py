"""Host post-processing of the Pix2Poly predictor (a-12 tail; predict/predictor_pix2poly.py): token sequences -> vertex coordinates,
permutation matrices -> polygons.  Plain Python / torch on whatever device the inputs live on: a few hundred vertices per tile, no kernel.

    postprocess(batch_preds, tokenizer)                 predictor_pix2poly.py:284-305  (EOS search + the (EOS-1) % 2 sanity check)
    permutations_to_polygons(perm, graph, out)          :213-282   defined for permutation matrices (what scores_to_permutations returns)
    coord_and_perm_to_polygons(coord_preds, perm_preds, tokenizer, max_num_vertices)      :111-138
    batch_to_polygons(model, tokenizer, x_images, x_lidar)      :141-152, the whole predict tail on the HIP path:
        encoder -> KV-cached greedy decode -> ScoreNets + device Hungarian -> polygons
"""
import numpy as np
import torch

from .pix2poly import scores_to_permutations


def postprocess(batch_preds, tokenizer):
    """-> per tile: ndarray [n, 2] of de-quantised (row, col) vertex coordinates, or None when the sequence fails the EOS check."""
    eos = (batch_preds == tokenizer.EOS_code).float().argmax(dim=-1)
    eos = torch.where((eos - 1) % tokenizer.token_mode != 0, torch.zeros_like(eos), eos)
    out = []
    for i, e in enumerate(eos.tolist()):
        out.append(None if e == 0 else tokenizer.decode(batch_preds[i, :e + 1]))
    return out


def _cycles(succ):
    """cycles of a permutation given as successor list, each starting at its smallest element and closed by repeating it; ordered by
    that smallest element (the order the reference's pairwise merging produces)"""
    seen, out = [False] * len(succ), []
    for s in range(len(succ)):
        if seen[s]:
            continue
        cyc, k = [s], succ[s]
        seen[s] = True
        while k != s and not seen[k]:
            seen[k] = True
            cyc.append(k)
            k = succ[k]
        cyc.append(k)
        out.append(cyc)
    return out


def permutations_to_polygons(perm, graph, out="torch"):
    """perm [B,N,N] 0/1 permutation matrices, graph: per tile [N,2] vertex coordinates -> per tile a list of closed polygons
    (first vertex repeated at the end).  Vertices mapped to themselves (diagonal 1) are padding and dropped."""
    B, N, _ = perm.shape
    batch = []
    for b in range(B):
        p = perm[b]
        idx = torch.nonzero(p.diagonal() == 0).view(-1)
        if idx.numel() == 0:
            batch.append([])
            continue
        sub = p[idx][:, idx]
        succ = torch.argmax(sub, dim=1).tolist()
        g = graph[b][idx.to(graph[b].device), :]
        polys = []
        for cyc in _cycles(succ):
            pts = g[cyc, :]
            if out in ("torch", "inria-torch"):
                polys.append(pts)
            elif out == "numpy":
                polys.append(pts.cpu().numpy())
            elif out == "list":
                q = pts * 300 / 320
                q[:, 0] = -q[:, 0]
                polys.append(torch.fliplr(q).tolist())
            elif out == "coco":
                polys.append(torch.fliplr(pts).reshape(-1).tolist())
            else:
                raise ValueError(f"unknown polygon format {out!r}: torch | numpy | list | coco | inria-torch")
        batch.append(polys)
    return batch


def coord_and_perm_to_polygons(coord_preds, perm_preds, tokenizer, max_num_vertices=None):
    """token sequences [B,L] + permutation matrices [B,N,N] -> per tile a list of [n,2] (x, y) polygons in pixels."""
    nv = max_num_vertices if max_num_vertices is not None else perm_preds.shape[-1]
    pad = float(tokenizer.PAD_code)
    coords = []
    for c in postprocess(coord_preds, tokenizer):
        c = torch.from_numpy(np.asarray(c)) if c is not None else torch.zeros((0, 2))
        c = c.reshape(-1, 2).to(torch.float32) if c.numel() else torch.zeros((0, 2))
        coords.append(torch.cat([c, torch.full((nv - c.shape[0], 2), pad)], dim=0))
    out = []
    for polys in permutations_to_polygons(perm_preds.cpu(), coords, out="torch"):
        keep = []
        for p in polys:
            p = torch.fliplr(p)
            p = p[p[:, 0] != pad]
            if len(p) > 0:
                keep.append(p)
        out.append(keep)
    return out


@torch.no_grad()
def batch_to_polygons(model, tokenizer, x_images=None, x_lidar=None, graphs=False):
    """One batch of tiles -> polygons per tile (Predictor.batch_to_polygons): every device stage on the HIP path."""
    enc = model.cfg.experiment.encoder
    if enc.use_images and enc.use_lidar:
        feats = model.encoder(x_images, x_lidar)
    elif enc.use_images:
        feats = model.encoder(x_images)
    elif enc.use_lidar:
        feats = model.encoder(x_lidar)
    else:
        raise ValueError("At least one of use_images or use_lidar must be True")
    # generate() returns the pre-logit features of the 385 decoded positions: the tensor the reference's LAST full `predict` pass
    # returns (the decoder is causal), which is what it feeds both ScoreNets (predictor_pix2poly.py:204-207)
    tokens, dec_feats = model.generate(feats, graphs=graphs)
    perm = scores_to_permutations(model.perm_scores(dec_feats))
    return coord_and_perm_to_polygons(tokens.cpu(), perm.cpu(), tokenizer, model.max_num_vertices)
