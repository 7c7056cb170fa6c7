"""Builds libp3hip.so (all HIP kernels + the C-ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libp3hip.so")
OBJ = os.path.join(CSRC, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _digest(path):
    h = hashlib.sha1()
    for dep in [path] + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + \
            [os.path.join(os.path.dirname(HERE), "include", "p3hip.h")]:
        with open(dep, "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src + ".o")
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = ["hipcc"] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as fh:
        fh.write(dig)
    return obj, True


def build_library(verbose=True, jobs=6):
    os.makedirs(OBJ, exist_ok=True)
    srcs = _sources()
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(_compile, srcs))
    objs = [o for o, _ in res]
    changed = any(c for _, c in res) or not os.path.exists(OUT)
    if changed:
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    if verbose:
        print(f"[p3hip] {'built' if changed else 'up to date'}: {OUT} ({len(srcs)} sources)")
    return OUT


if __name__ == "__main__":
    build_library()
