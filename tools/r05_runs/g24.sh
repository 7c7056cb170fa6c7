# r05: priority skew variants of pair_bwd_x3 (P3_PX_MODE 1 default lib, 0 / 2 / 3 variant libs): same-box kernel times
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x -k "pair_bwd_fused_x3" 2>&1 | tail -2
for L in libp3hip.so libp3hip_m0.so libp3hip_m2.so libp3hip_m3.so; do
  P3HIP_LIB=$GRAFT_REPO_ROOT/pixelspointspolygons_amd/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$L -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g24_$L.log 2>&1
  find /tmp/pf_$L -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g24_$L.csv \;
  echo "== $L"; python tools/kstats.py gpurun_out/r05/g24_$L.csv 13 70 | grep -E "total|pair_bwd_x3" | cut -c1-120
done
