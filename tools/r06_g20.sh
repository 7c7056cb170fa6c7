#!/bin/bash
# r06 g20: full GPU suite (A-stationary default rule without the planes + multiplier epilogue, wide weight-gradient tile, write-through stores)
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_gpu_suite_run2.txt
cat gpurun_out/r06_gpu_suite_run2.txt
