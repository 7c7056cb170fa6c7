"""Which parameter gradients / outputs of one early-fusion train step differ between two runs of the same process state (bitwise)?
python tools/diag_determinism.py [precision]      (fp32x3 default; FlatAdamW direct gradients like the bench)"""
import sys
import torch
sys.path.insert(0, ".")
from oracle import p3_oracle as O      # tools/ is test infrastructure like tests/: seeded inputs / weights only
from pixelspointspolygons_amd import ops
from pixelspointspolygons_amd.config import make_config
from pixelspointspolygons_amd.pix2poly import Pix2PolyModel, Tokenizer
from pixelspointspolygons_amd.training import FlatAdamW, pix2poly_loss

prec = sys.argv[1] if len(sys.argv) > 1 else "fp32x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sd = O.make_state_dict("fusion", seed=42)
inp = {k: v.cuda() for k, v in O.make_inputs(B, seed=5).items()}


def run():
    ops.reset_process_state()
    cfg = make_config("early_fusion_vit", precision=prec, device="cuda")
    m = Pix2PolyModel(cfg, Tokenizer(cfg).vocab_size, 0)
    m.load_state_dict(sd, strict=True)
    m.train()
    m.decoder.set_dropout(0.0)
    opt = FlatAdamW(m, compute_dtype=torch.float32 if prec != "bf16" else torch.bfloat16)
    opt.zero_grad()
    logits, perm = m(inp["image"], (inp["lidar_values"], inp["lidar_offsets"]), inp["y"][:, :-1])
    loss = pix2poly_loss(logits, perm, inp["y"][:, 1:], inp["y_perm"])[0]
    loss.backward()
    torch.cuda.synchronize()
    out = {"loss": loss.detach().clone(), "logits": logits.detach().clone(), "perm": perm.detach().clone()}
    out.update({"g:" + k: p.grad.detach().clone() for k, p in m.named_parameters()})
    out.update({"b:" + k: v.detach().clone() for k, v in m.named_buffers() if v.is_floating_point()})
    opt.close()
    return out


a, b = run(), run()
bad = [(k, float((a[k].float() - b[k].float()).abs().max() / a[k].float().abs().max().clamp_min(1e-30))) for k in a if not torch.equal(a[k], b[k])]
print(f"[{prec}] {len(bad)} of {len(a)} tensors differ between two runs")
for k, e in sorted(bad, key=lambda kv: -kv[1]):
    print(f"   {e:.2e}  {k}")
