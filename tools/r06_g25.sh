#!/bin/bash
# r06 g25: (1) the conv weight gradients' nine products parked + one flush, step same-box; (2) timing ablations of the fp32x3 attention kernels (staging split / P split /
# exponentials removed - wrong numbers, what each costs); (3) the per-shape GEMM table of the fp32x3 step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_g25.txt
: > $O
timeout 600 python -m pytest tests/test_train_gpu.py tests/test_ffl_gpu.py -q -m gpu -x 2>&1 | tail -3 | tee -a $O
for i in 1 2 3; do
  for P in 1 0; do
    echo -n "conv_park=$P " >> $O
    P3_CONV_PARK=$P timeout 300 python bench.py --lean --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['final_loss'])" >> $O
  done
done
for L in base25 attnabl1 attnabl2 attnabl4 attnabl7 base25; do
  echo -n "$L " >> $O
  P3HIP_LIB=tmp_ab/libp3hip_$L.so timeout 300 python tools/mb_attn_x3.py 2>&1 | tail -1 >> $O
done
cat $O
timeout 600 python tools/gemm_shape_table.py --precision fp32x3 > gpurun_out/r06_shapes_fp32x3.txt 2>&1
tail -3 gpurun_out/r06_shapes_fp32x3.txt
