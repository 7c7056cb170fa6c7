set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "sinkhorn or gemm_tn" 2>&1 | tail -40 > gpurun_out/r04/t6.log
python tools/mb_sinkhorn.py > gpurun_out/r04/mb_sinkhorn2.txt 2>&1
python tools/mb_tn.py > gpurun_out/r04/mb_tn_dma.txt 2>&1
P3_TN_DMA=0 python tools/mb_tn.py > gpurun_out/r04/mb_tn_old.txt 2>&1
P3_TN_DMA_NBUF=3 python tools/mb_tn.py > gpurun_out/r04/mb_tn_dma_nbuf3.txt 2>&1
python -X faulthandler -m pytest tests/test_pillar_membership_gpu.py tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r04/t1c.log 2>&1
head -120 gpurun_out/r04/t1c.log
cat gpurun_out/r04/t6.log gpurun_out/r04/mb_sinkhorn2.txt gpurun_out/r04/mb_tn_dma.txt gpurun_out/r04/mb_tn_old.txt gpurun_out/r04/mb_tn_dma_nbuf3.txt
