set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_pillar_membership_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r04/t1.log
python -m pytest tests/test_ffl_gpu.py -x -q -m gpu -k full_batch 2>&1 | tail -15 > gpurun_out/r04/t2.log
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "twin or carrier or stream" 2>&1 | tail -15 > gpurun_out/r04/t3.log
P3_SIDE_SN=1 P3_SIDE_DW=1 P3_SIDE_STEM=1 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py -x -q -m gpu -k "train_step or overfit" 2>&1 | tail -15 > gpurun_out/r04/t4.log
bash tools/streams_ab.sh 3 > gpurun_out/r04/streams_ab.txt 2>&1
python bench.py > gpurun_out/r04/bench_full.log 2>&1
tail -1 gpurun_out/r04/bench_full.log > gpurun_out/r04/bench0.json
cat gpurun_out/r04/t1.log gpurun_out/r04/t2.log gpurun_out/r04/t3.log gpurun_out/r04/t4.log gpurun_out/r04/streams_ab.txt
cut -c1-1500 gpurun_out/r04/bench0.json
