"""ctypes loader for libp3hip.so — the only compute backend of this package.

There is deliberately NO fallback: if the HIP extension is missing or a symbol is absent the import
of any op fails loudly (RuntimeError), so a silent eager/PyTorch path can never masquerade as the product.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# P3HIP_LIB selects another build of the same library (A/B runs of two kernel variants on one box: tools/ab.sh)
LIB_PATH = os.environ.get("P3HIP_LIB") or os.path.join(_HERE, "libp3hip.so")
_lib = None


class P3Error(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise P3Error(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU / eager fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.p3_last_error_string.restype = ctypes.c_char_p
        _lib.p3_version.restype = ctypes.c_int
        _lib.p3_last_kernel.restype = ctypes.c_char_p
        _lib.p3_trace_kernels.restype = None
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().p3_last_error_string().decode()
        raise P3Error(f"{what} failed with code {rc}: {msg}")
