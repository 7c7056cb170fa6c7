cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_d40 -o st -- python bench.py --lean --points 40000 --steps 10 > gpurun_out/r04/d40_run.log 2>&1
tail -1 gpurun_out/r04/d40_run.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('40k points ms/step', d['ms_per_step'])"
find /tmp/pf_d40 -name "*kernel_stats.csv" -exec cp {} gpurun_out/r04/r04_dense40k_step_kernel_stats.csv \;
python tools/kstats.py gpurun_out/r04/r04_dense40k_step_kernel_stats.csv 15 200 | grep -E "pfn|pillar|total|scatter|assemble" | head -30
