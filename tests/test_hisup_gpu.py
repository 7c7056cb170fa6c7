"""HiSup head set on the HIP path against the REFERENCE'S OWN module: tests/golden/hisup_heads.npz was produced by
importing /root/reference/pixelspointspolygons/models/hisup/model_hisup.py (EncoderDecoder.forward_common over a pass-through encoder,
tests/golden/make_hisup_heads_golden.py) in eval mode and in train mode, with the 17 BatchNorm running-statistic updates."""
import pytest
import torch

from tests.helpers import load_golden, rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
HEADS = ("joff", "jloc", "mask", "afm", "remask")


def _model(weights, precision):
    from pixelspointspolygons_amd.hisup import HiSupHeads
    m = HiSupHeads(dim_in=32, precision=precision)
    m.load_state_dict(weights, strict=True)            # the reference's parameter / buffer names
    return m.to(DEV)


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_hisup_heads_eval_vs_reference_module(precision, tol):
    d, w = load_golden("hisup_heads.npz")
    m = _model(w, precision).eval()
    out = m(d["features"].to(DEV))
    for k in HEADS:
        assert out[k].shape == d["eval." + k].shape and out[k].dtype == torch.float32
        assert rel_err(out[k].cpu(), d["eval." + k]) < tol, (k, rel_err(out[k].cpu(), d["eval." + k]))
    for name, buf in m.named_buffers():                # eval mode leaves every BatchNorm buffer alone
        assert torch.equal(buf.cpu(), w[name]), name


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-3), ("bf16", 6e-2)])
def test_hisup_heads_train_mode_batch_statistics_and_running_stats(precision, tol):
    d, w = load_golden("hisup_heads.npz")
    m = _model(w, precision).train()
    out = m(d["features"].to(DEV))
    for k in HEADS:
        assert rel_err(out[k].cpu(), d["train." + k]) < tol, (k, rel_err(out[k].cpu(), d["train." + k]))
    after = {k[len("after."):]: v for k, v in d.items() if k.startswith("after.")}
    assert len([k for k in after if k.endswith("num_batches_tracked")]) == 17
    for name, buf in m.named_buffers():
        ref = after[name]
        if name.endswith("num_batches_tracked"):
            assert int(buf) == int(ref), name
        else:
            assert rel_err(buf.cpu(), ref) < (1e-4 if precision == "fp32" else 2e-2), (name, rel_err(buf.cpu(), ref))


def test_hisup_heads_at_the_configured_width_matches_the_oracle():
    """in_feature_dim = 256 (config/model/hisup.yaml) on a 56 x 56 map: the HIP path against the oracle restatement (itself pinned against the
    reference module by tests/test_oracle_golden.py) - the golden fixture is 32 channels wide to stay small."""
    from oracle import p3_oracle as O
    from pixelspointspolygons_amd.hisup import HiSupHeads
    torch.manual_seed(7)
    m = HiSupHeads(dim_in=256, precision="fp32")
    gen = torch.Generator().manual_seed(8)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
            mod.running_var.copy_(torch.rand(mod.num_features, generator=gen) + 0.5)
            mod.weight.data.copy_(torch.rand(mod.num_features, generator=gen) + 0.5)
            mod.bias.data.copy_(torch.randn(mod.num_features, generator=gen) * 0.1)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    feats = torch.randn(2, 256, 56, 56, generator=gen)
    for training in (False, True):
        sd_ref = {k: v.clone() for k, v in sd.items()}
        with torch.no_grad():
            ref = O.hisup_heads(feats, sd_ref, training=training)
        m.load_state_dict(sd)
        m = m.to(DEV).train(training)
        out = m(feats.to(DEV))
        for k in HEADS:
            assert rel_err(out[k].cpu(), ref[k]) < 1e-3, (training, k, rel_err(out[k].cpu(), ref[k]))
        if training:
            for name, buf in m.named_buffers():
                if not name.endswith("num_batches_tracked"):
                    assert rel_err(buf.cpu(), sd_ref[name]) < 1e-4, name
        m = m.cpu()


def test_hisup_heads_reject_host_tensors_and_wrong_width():
    from pixelspointspolygons_amd import hip
    from pixelspointspolygons_amd.hisup import HiSupHeads
    m = HiSupHeads(dim_in=32, precision="fp32").to(DEV).eval()
    with pytest.raises(hip.P3Error):
        m(torch.zeros(1, 32, 8, 8))
    with pytest.raises(hip.P3Error):
        m(torch.zeros(1, 64, 8, 8, device=DEV))
