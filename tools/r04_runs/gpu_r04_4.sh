set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_ops_gpu.py tests/test_backward_gpu.py -x -q -m gpu -k "attention" 2>&1 | tail -5 > gpurun_out/r04/t7.log
for i in 1 2; do
  echo "== two sets (default)"; python tools/mb_attn.py 2>&1 | grep shape
  echo "== one set"; P3HIP_LIB=$PWD/tmp_ab/pf1.so python tools/mb_attn.py 2>&1 | grep shape
  echo "== two sets, pair-major order"; P3_ATTN_TAIL_FIRST=0 python tools/mb_attn.py 2>&1 | grep shape
done > gpurun_out/r04/mb_attn_ab.txt 2>&1
for i in 1 2; do
for L in "" "$PWD/tmp_ab/pf1.so"; do
  echo -n "lib=${L:-default}: "; P3HIP_LIB=$L python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
done; done > gpurun_out/r04/attn_step_ab.txt 2>&1
cat gpurun_out/r04/t7.log gpurun_out/r04/mb_attn_ab.txt gpurun_out/r04/attn_step_ab.txt
