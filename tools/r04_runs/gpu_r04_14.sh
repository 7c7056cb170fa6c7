cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python tools/gemm_shape_table.py --lean > gpurun_out/r04/gemm_shape_table.txt 2>&1
cat gpurun_out/r04/gemm_shape_table.txt | tail -80
