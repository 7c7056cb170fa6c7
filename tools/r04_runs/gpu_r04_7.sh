cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "pair_bwd_fused or scorenet or train_step" 2>&1 | tail -8
python - <<'PY'
import sys, torch
sys.path.insert(0, ".")
import pixelspointspolygons_amd.hip as h
from tools.microbench import timeit
B, N = 64, 192
R = B * N * N
dH2 = (torch.randn(R, 128, device="cuda") * 0.5).bfloat16()
w2t = (torch.randn(256, 128, device="cuda") * 0.1).bfloat16()
U, V = torch.randn(B * N, 256, device="cuda").bfloat16(), torch.randn(B * N, 256, device="cuda").bfloat16()
sc, sh, mu = torch.rand(256, device="cuda") + 0.5, torch.randn(256, device="cuda") * 0.3, torch.randn(256, device="cuda")
acc = torch.zeros(512, device="cuda")
tf = timeit(lambda: h.pair_bwd_fused(dH2, w2t, U, V, sc, sh, mu, B, N, acc))
def two():
    dA2 = h.gemm(dH2, w2t, out_dtype=torch.bfloat16)
    return h.pair_bwd(dA2, U, V, sc, sh, mu, B, N, acc)
t2 = timeit(two)
print(f"pair backward at B=64, N=192: fused {tf*1e6:.1f} us, gemm + pair_bwd {t2*1e6:.1f} us")
PY
for i in 1 2; do python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['final_loss'])"; done
