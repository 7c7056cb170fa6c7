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


def test_descriptor_structs_have_the_layout_the_header_declares(tmp_path):
    """The ctypes mirrors in hip.py (GemmDesc, AttnDesc, ...) must match include/p3hip.h field for field: a probe compiled with gcc from
    the header itself prints sizeof / offsetof of every member, compared with the ctypes classes (names, order, offsets, total size)."""
    import shutil
    import subprocess
    import pixelspointspolygons_amd.hip as h
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    mirrors = {"p3_dropout": h.Dropout, "p3_gemm_desc": h.GemmDesc, "p3_attn_desc": h.AttnDesc, "p3_pillar_desc": h.PillarDesc,
               "p3_decode_layer_desc": h.DecodeLayerDesc, "p3_gemm_x3_desc": h.GemmX3Desc}
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(p3_\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            for part in decl.split(","):               # "int M, N, K" / "const float *g1, *be1"
                name = re.findall(r"(\w+)\s*(?:\[[^\]]*\])?\s*$", part.strip())
                assert name, decl
                fields.append(name[0])
        structs[m.group(2)] = fields
    assert set(mirrors) == set(structs), (sorted(structs), sorted(mirrors))
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', 'int main(void) {']
    for sname, fields in structs.items():
        src.append(f'  printf("{sname} size %zu\\n", sizeof({sname}));')
        for f in fields:
            src.append(f'  printf("{sname} {f} %zu\\n", offsetof({sname}, {f}));')
    src += ['  return 0;', '}']
    c = tmp_path / "probe.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(c)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {}
    for line in out.splitlines():
        sname, f, v = line.split()
        got.setdefault(sname, []).append((f, int(v)))
    for sname, cls in mirrors.items():
        rows = got[sname]
        assert rows[0] == ("size", ctypes.sizeof(cls)), (sname, rows[0], ctypes.sizeof(cls))
        want = [(n, getattr(cls, n).offset) for n, *_ in cls._fields_]
        assert rows[1:] == want, (sname, [a for a, b in zip(rows[1:], want) if a != b][:4])


def _split_top_level(argtext):
    """split on commas that are not nested in (), [] or {}"""
    out, depth, cur = [], 0, []
    for ch in argtext:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append("".join(cur)); cur = []
        else:
            cur.append(ch)
    if "".join(cur).strip():
        out.append("".join(cur))
    return out


def test_every_binding_call_passes_as_many_arguments_as_the_header_declares():
    """ctypes does not check arity: a call that drifted from its prototype would read garbage registers on the device side.  Static check of
    every `lib().p3_*(...)` call in the package against the parameter count of the declaration in include/p3hip.h."""
    text = re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(p3_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        protos[m.group(1)] = 0 if args in ("", "void") else len(_split_top_level(args))
    checked = 0
    pkg = os.path.join(ROOT, "pixelspointspolygons_amd")
    for fn in sorted(os.listdir(pkg)):
        if not fn.endswith(".py"):
            continue
        src = open(os.path.join(pkg, fn)).read()
        for m in re.finditer(r"(?:lib\(\)|\bL)\.(p3_\w+)\(", src):
            name, i, depth = m.group(1), m.end(), 1
            j = i
            while depth:
                depth += {"(": 1, ")": -1}.get(src[j], 0)
                j += 1
            parts = _split_top_level(src[i:j - 1])
            assert name in protos, (fn, name)
            if any(a.strip().startswith("*") for a in parts):       # star-expanded argument tuples (the pillar stem variants) cannot be counted statically
                continue
            nargs = len(parts)
            assert nargs == protos[name], (fn, name, nargs, protos[name])
            checked += 1
    assert checked >= 55
