# r05: the default bench line again (the weight-gradient kernel's timer label now is the name rocprofv3 prints)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python bench.py 2>&1 | tail -1 > gpurun_out/final/r05_bench.json
cut -c1-400 gpurun_out/final/r05_bench.json
