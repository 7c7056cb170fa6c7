"""Stage timestamps of one plain bf16 GEMM tile (workgroup 0): needs a library built with -DP3_GEMM_TIMING, e.g.
   tools/build_variant.sh tmp_ab/gemm_timing.so gemm.hip -DP3_GEMM_TIMING ; P3HIP_LIB=tmp_ab/gemm_timing.so python tools/mb_gemm_stages.py [M N K]"""
import sys
import torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h

M, N, K = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (64 * 785, 1152, 384)
a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16(); b = torch.randn(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
ts = torch.zeros(16, dtype=torch.int64, device="cuda")
lib = h.lib()
real = lib.p3_gemm
def patched(A, W, C, dref, s):                 # the timestamp buffer travels in the (unused in plain mode) pair_V field
    dref._obj.pair_V = ts.data_ptr()
    return real(A, W, C, dref, s)
lib.p3_gemm = patched
for _ in range(5):
    h.gemm(a, w, bias=b, out=out)
torch.cuda.synchronize()
t = ts.cpu().tolist()
for i, name in enumerate(["row sources + kernel args", "prologue (2 slices loaded, 1 stored)", "K loop", "epilogue"], 1):
    print(f"{name:38s} {(t[i] - t[i - 1]) * 10:6d} ns")
print("tile", (t[4] - t[0]) * 10, "ns")
