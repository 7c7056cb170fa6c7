# r05 first lease: fp32x3 baseline on this round's box: per-shape GEMM table, lean bench line, rocprof kernel stats
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python tools/gemm_shape_table.py --precision fp32x3 > gpurun_out/r05/g01_shapes_fp32x3.txt 2>&1
python bench.py --lean --precision fp32x3 2>&1 | tail -1 > gpurun_out/r05/g01_bench_fp32x3.json
python bench.py --lean --precision bf16 2>&1 | tail -1 > gpurun_out/r05/g01_bench_bf16.json
tail -c 600 gpurun_out/r05/g01_bench_fp32x3.json
head -40 gpurun_out/r05/g01_shapes_fp32x3.txt
