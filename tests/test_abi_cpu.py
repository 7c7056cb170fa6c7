"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports every symbol include/p3hip.h declares;
argument validation (which runs before any HIP call) returns the documented negative codes."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "p3hip.h")


def _declared():
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(p3_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from pixelspointspolygons_amd.build import build_library
    so = build_library(verbose=False)
    return ctypes.CDLL(so)


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("p3_gemm", "p3_attention", "p3_attention_bwd", "p3_layernorm", "p3_pillar_stem", "p3_sinkhorn", "p3_sinkhorn_bwd",
                 "p3_gemm_tn", "p3_adamw", "p3_ce_loss_fwd", "p3_score_out", "p3_last_error_string", "p3_version"):
        assert must in names
    assert len(names) >= 40


def test_library_exports_every_declared_symbol(lib):
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_version_and_error_convention_without_gpu(lib):
    lib.p3_version.restype = ctypes.c_int
    lib.p3_last_error_string.restype = ctypes.c_char_p
    assert lib.p3_version() >= 100
    # null pointers are rejected before any device work: P3_EINVAL (-1) + message
    rc = lib.p3_gemm(None, None, None, None, None)
    assert rc == -1 and b"p3_gemm" in lib.p3_last_error_string()
    rc = lib.p3_sinkhorn(None, None, 0, 0, 0, 0, None, None, None, None)
    assert rc == -1


def test_product_fails_loudly_on_host_tensors():
    """There is no CPU fallback: host tensors are refused instead of being routed to eager PyTorch."""
    import torch
    import pixelspointspolygons_amd.hip as h
    from pixelspointspolygons_amd._lib import P3Error
    with pytest.raises(P3Error):
        h.gemm(torch.zeros(8, 64), torch.zeros(8, 64))
    with pytest.raises(P3Error):
        h.attention(torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), torch.zeros(1, 4, 64), 1, 1.0)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pixelspointspolygons_amd/ may import, load or call it."""
    pkg = os.path.join(ROOT, "pixelspointspolygons_amd")
    pat = re.compile(r"^\s*(from|import)\s+[^#\n]*oracle|p3_oracle|libp3oracle|/root/reference", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not pat.search(src), (f, pat.search(src).group(0))
