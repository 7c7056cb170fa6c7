cd $GRAFT_REPO_ROOT
T0=$(date +%s)
python bench.py > /tmp/b.json 2> /tmp/b.err
T1=$(date +%s)
echo "bench.py default run: $((T1-T0)) s wall, exit $?"
tail -2 /tmp/b.err
python -c "import json; d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], sorted(k for k in d.keys()))"
