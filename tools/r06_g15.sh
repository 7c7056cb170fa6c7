#!/bin/bash
# r06 g15: full GPU suite on the A-stationary default rule, then the lean bench line (step time) in fp32x3
mkdir -p gpurun_out
timeout 2700 python -m pytest tests -q -m gpu -x 2>&1 | tail -15 > gpurun_out/r06_gpu_suite_run1.txt
cat gpurun_out/r06_gpu_suite_run1.txt
timeout 600 python bench.py --lean --steps 10 2>&1 | tail -1 > gpurun_out/r06_bench_lean_1.json
cut -c1-400 gpurun_out/r06_bench_lean_1.json
