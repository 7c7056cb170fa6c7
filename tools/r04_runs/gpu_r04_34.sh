cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 600 python tools/aten_sites.py > gpurun_out/r04/aten_sites.txt 2>&1
tail -75 gpurun_out/r04/aten_sites.txt
