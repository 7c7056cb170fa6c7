#!/bin/bash
# r06 g21: the new full-length greedy test, then the round's artefacts (tools/final_prof_r06.sh)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -k "all_385_steps" -s 2>&1 | tail -8 > gpurun_out/r06_greedy385.txt
cat gpurun_out/r06_greedy385.txt
bash tools/final_prof_r06.sh > gpurun_out/final_prof_r06.log 2>&1
tail -30 gpurun_out/final_prof_r06.log | cut -c1-900
