#!/bin/bash
# A/B two builds of libp3hip.so on ONE box (bench noise between boxes is ~1 %): tools/ab.sh <libA.so> <libB.so> [bench args...]
# Build variant A into a copy first:  cp pixelspointspolygons_amd/libp3hip.so /tmp/a.so  (the .so must live under the repo to travel)
A=$1; B=$2; shift 2
for i in 1 2 3; do
  for L in $A $B; do
    echo -n "$(basename $L): "
    P3HIP_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fwd "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
