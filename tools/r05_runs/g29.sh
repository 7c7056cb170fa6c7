# r05: rows in flight per lane group in the rows8 pillar kernels (P3_PFN_NR = 8 default lib, 4 / 16 variant libs): 40 k-point kernel times, same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_backward_gpu.py tests/test_model_gpu.py -x -q -k "pillar or canvas or stem" 2>&1 | tail -2
for L in libp3hip.so libp3hip_nr4.so libp3hip_nr16.so; do
  P3HIP_LIB=$GRAFT_REPO_ROOT/pixelspointspolygons_amd/$L rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$L -o st -- python bench.py --lean --points 40000 --steps 6 --warmup 2 > gpurun_out/r05/g29_$L.log 2>&1
  find /tmp/pf_$L -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g29_$L.csv \;
  echo "== $L"; python tools/kstats.py gpurun_out/r05/g29_$L.csv 9 70 | grep -E "total|rows8|reduce8"
done
