#!/bin/bash
# r06 g14: A-stationary kernel v9 (stores counted into the vmcnt allowance of the DMA waves): check, timing; then the x3 GPU tests and the model tests on the new default rule
mkdir -p gpurun_out
O=gpurun_out/mb_as_14.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
grep -v amdgpu.ids $O | tail -14
timeout 1500 python -m pytest tests/test_x3_gpu.py tests/test_model_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O
