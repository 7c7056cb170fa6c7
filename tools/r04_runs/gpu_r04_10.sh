cd $GRAFT_REPO_ROOT
python -m pytest tests/test_backward_gpu.py -x -q -m gpu -k "pair_node or bn_sums or scorenet" 2>&1 | tail -5
python -m pytest tests/test_syncbn_gpu.py tests/test_input_pipeline_gpu.py tests/test_rccl_single_rank_gpu.py -x -q -m gpu 2>&1 | tail -5
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "kv_cached or greedy" -s 2>&1 | grep -E "passed|failed|fused decode|Error|error" | tail -8
python bench.py --lean --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms/step', d['ms_per_step'], d['final_loss'])"
