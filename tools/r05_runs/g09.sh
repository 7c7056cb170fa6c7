# r05 lease 9: full GPU suite + the default bench line (fp32x3 headline) on one box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 3000 python -m pytest tests/ -q -m gpu > gpurun_out/r05/g09_suite.txt 2>&1
echo "pytest exit $?" >> gpurun_out/r05/g09_suite.txt
grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05/g09_suite.txt | head -40
timeout 1500 python bench.py > gpurun_out/r05/g09_bench_stdout.txt 2> gpurun_out/r05/g09_bench_stderr.txt
tail -1 gpurun_out/r05/g09_bench_stdout.txt > gpurun_out/r05/g09_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r05/g09_bench.json'))
print('value', d['value'], d['ms_per_step'], d['dtype'])
print('roofline', json.dumps(d['roofline'])[:600])
for k in d:
    if k.endswith('_vs_oracle') or k.endswith('_mode'): print(k, json.dumps(d[k])[:300])
"
tail -5 gpurun_out/r05/g09_bench_stderr.txt
