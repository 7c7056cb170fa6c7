# r05: s_setprio skew of the two waves of a SIMD in pair_bwd_x3 (waves 0..3 own the matrix pipe during their product phase): same-box A/B of the kernel time
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
timeout 600 python -m pytest tests/test_backward_gpu.py -q -x -k "pair_bwd_fused_x3" 2>&1 | tail -2
for L in libp3hip.so libp3hip_prio0.so; do
  P3HIP_LIB=$GRAFT_REPO_ROOT/pixelspointspolygons_amd/$L timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pf_$L -o st -- python bench.py --lean --steps 10 --warmup 3 > gpurun_out/r05/g23_$L.log 2>&1
  find /tmp/pf_$L -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/g23_$L.csv \;
  echo "== $L"; python tools/kstats.py gpurun_out/r05/g23_$L.csv 13 70 | grep -E "total|pair_bwd_x3|pair_fwd_x3|pair_dw_x3" | cut -c1-120
done
