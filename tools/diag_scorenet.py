"""Diagnostic (GPU): where the ScoreNet backward's error against float64 comes from at N = 192 - arithmetic or ReLU-kink flips.

Runs the HIP forward + backward several times (run-to-run spread = reduction order), counts the ReLU decisions that differ from the
float64 restatement and re-evaluates the float64 gradient with the product's own decisions replayed ("arithmetic-only" error).
Test infrastructure: imports the oracle.  Usage: python tools/diag_scorenet.py [runs] [B] [N]
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import p3_oracle as O  # noqa: E402


def _rand(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def staged64(sd, feats, g, B, N, masks=None, transpose=False):
    """float64 restatement in the product's staging (U, V, H2, H3); masks = optional (m1, m2, m3) overrides of the ReLU decisions."""
    dt = torch.float64
    W = {k[len("scorenet1."):]: v.to(dt) for k, v in sd.items() if k.startswith("scorenet1.")}
    D, eps = 256, 1e-5
    F = feats[:, 1:].reshape(B, N, 2, D).to(dt).mean(2).reshape(B * N, D)
    W1 = W["conv1.weight"].reshape(256, 512)
    U = F @ W1[:, :D].t() + W["conv1.bias"]
    V = F @ W1[:, D:].t()
    P = (U.view(B, N, 1, 256) + V.view(B, 1, N, 256)).reshape(-1, 256)
    R = P.shape[0]

    def bn(H, pre):
        m = H.mean(0)
        var = H.var(0, unbiased=False)
        rs = 1 / torch.sqrt(var + eps)
        z = (H - m) * rs * W[pre + ".weight"] + W[pre + ".bias"]
        return z, m, rs
    z1, m1, r1 = bn(P, "bn1")
    k1 = (z1 > 0) if masks is None else masks[0]
    A1 = z1 * k1
    H2 = A1 @ W["conv2.weight"].reshape(128, 256).t() + W["conv2.bias"]
    z2, m2, r2 = bn(H2, "bn2")
    k2 = (z2 > 0) if masks is None else masks[1]
    A2 = z2 * k2
    H3 = A2 @ W["conv3.weight"].reshape(64, 128).t() + W["conv3.bias"]
    z3, m3, r3 = bn(H3, "bn3")
    k3 = (z3 > 0) if masks is None else masks[2]
    A3 = z3 * k3
    w4 = W["conv4.weight"].reshape(64)
    dS = (g.transpose(1, 2) if transpose else g).reshape(-1).to(dt)

    def bn_bwd(G, H, k, m, rs, gamma):
        dz = G * k
        xh = (H - m) * rs
        dbeta, dgamma = dz.sum(0), (dz * xh).sum(0)
        dH = gamma * rs * (dz - dbeta / R - xh * dgamma / R)
        return dH, dgamma, dbeta
    G3 = dS[:, None] * w4[None, :]
    dw4, db4 = (dS[:, None] * A3).sum(0), dS.sum()
    dH3, dg3, dbt3 = bn_bwd(G3, H3, k3, m3, r3, W["bn3.weight"])
    dW3 = dH3.t() @ A2
    dA3 = dH3 @ W["conv3.weight"].reshape(64, 128)
    dH2, dg2, dbt2 = bn_bwd(dA3, H2, k2, m2, r2, W["bn2.weight"])
    dW2 = dH2.t() @ A1
    dA2 = dH2 @ W["conv2.weight"].reshape(128, 256)
    dH1, dg1, dbt1 = bn_bwd(dA2, P, k1, m1, r1, W["bn1.weight"])
    dU = dH1.view(B, N, N, 256).sum(2).reshape(B * N, 256)
    dV = dH1.view(B, N, N, 256).sum(1).reshape(B * N, 256)
    dW1 = torch.cat([dU.t() @ F, dV.t() @ F], 1)
    grads = {"conv1.weight": dW1.view(256, 512, 1, 1), "bn1.weight": dg1, "bn1.bias": dbt1, "conv2.weight": dW2.view(128, 256, 1, 1),
             "bn2.weight": dg2, "bn2.bias": dbt2, "conv3.weight": dW3.view(64, 128, 1, 1), "bn3.weight": dg3, "bn3.bias": dbt3,
             "conv4.weight": dw4.view(1, 64, 1, 1), "conv4.bias": db4.view(1)}
    return grads, (z1, z2, z3)


def main():
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 192
    from pixelspointspolygons_amd.pix2poly import ScoreNet, scorenet_forward
    from pixelspointspolygons_amd.backward import scorenet_backward
    dev = "cuda"
    out_rows = []
    for seed in (4, 14, 24):
        sd = O.make_state_dict("image", dict(dim=64, depth=1, heads=2, mlp=128, patch=8, img=32, eps=1e-6), seed=9, n_vertices=N)
        feats, g = _rand(B, 2 * N + 1, 256, seed=seed), _rand(B, N, N, seed=seed + 1)
        ref, zs = staged64(sd, feats, g, B, N)
        gn = max(float(v.norm()) for v in ref.values())
        net = ScoreNet(N, in_channels=512)
        net.load_state_dict({k[len("scorenet1."):]: v for k, v in sd.items() if k.startswith("scorenet1.")}, strict=True)
        net.cd = torch.float32
        net = net.to(dev).train(True)
        names = [n for n, _ in net.named_parameters()]
        prev = None
        for run in range(runs):
            net.train(True)
            fd = feats.to(dev)
            keep = {}
            out = torch.zeros(B, N, N, device=dev)
            with torch.no_grad():
                scorenet_forward(net, fd, out, False, keep)
                dfe, dparams = scorenet_backward(net, fd, keep, g.to(dev), False)
            torch.cuda.synchronize()
            got = {n: (t.detach().double().cpu() if t is not None else None) for n, t in zip(names, dparams)}
            # the product's own ReLU decisions from its saved state: sign(fma(H, scale, shift)) == sign of the exact value
            (sc1, sh1, _, _), (sc2, sh2, _, _), (sc3, sh3, _, _) = [tuple(t.double().cpu() for t in trip) for trip in keep["bn"]]
            U, V = keep["U"].double().cpu(), keep["V"].double().cpu()
            Pk = (U.view(B, N, 1, 256) + V.view(B, 1, N, 256)).float().double().reshape(-1, 256)      # the kernels add U + V in fp32
            k1 = (Pk * sc1 + sh1) > 0
            k2 = (keep["H2"].double().cpu() * sc2 + sh2) > 0
            k3 = (keep["H3"].double().cpu() * sc3 + sh3) > 0
            flips = [int((k != (z > 0)).sum()) for k, z in zip((k1, k2, k3), zs)]
            zmax = [float(z[k != (z > 0)].abs().max()) if f else 0.0 for k, z, f in zip((k1, k2, k3), zs, flips)]
            rep, _ = staged64(sd, feats, g, B, N, masks=(k1, k2, k3))
            row = {"seed": seed, "run": run, "flips": flips, "flip_zmax": zmax}
            for n in ("conv1.weight", "conv2.weight", "conv3.weight", "bn1.weight", "bn1.bias", "bn2.weight", "bn3.weight", "conv4.weight"):
                if got[n] is None:
                    continue
                e = float((got[n] - ref[n]).norm() / max(float(ref[n].norm()), 1e-3 * gn))
                er = float((got[n] - rep[n]).norm() / max(float(rep[n].norm()), 1e-3 * gn))
                row[n] = (e, er)
            row["bit_identical_to_prev_run"] = None if prev is None else all(
                torch.equal(got[n], prev[n]) for n in got if got[n] is not None)
            prev = got
            print(json.dumps(row), flush=True)
            out_rows.append(row)
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/diag_scorenet.json", "w") as f:
        json.dump(out_rows, f, indent=0)


if __name__ == "__main__":
    main()
