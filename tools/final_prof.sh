set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py 2>&1 | tail -1 > gpurun_out/final/bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_stats -o st -- python bench.py --no-cpu-baseline > gpurun_out/final/stats_run.log 2>&1
find /tmp/pf_stats -name "*kernel_stats.csv" -exec cp {} gpurun_out/final/kernel_stats.csv \;
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf_fetch -o f -- python bench.py --graph 0 --steps 4 --warmup 2 --no-cpu-baseline --no-fwd --no-kernel-timing --no-host-feed > gpurun_out/final/fetch_run.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pf_write -o w -- python bench.py --graph 0 --steps 4 --warmup 2 --no-cpu-baseline --no-fwd --no-kernel-timing --no-host-feed > gpurun_out/final/write_run.log 2>&1
python tools/pmc_traffic.py /tmp/pf_fetch /tmp/pf_write gpurun_out/final/pmc_traffic.json > gpurun_out/final/pmc_summary.txt 2>&1
tail -3 gpurun_out/final/pmc_summary.txt
cat gpurun_out/final/bench.json | cut -c1-400
