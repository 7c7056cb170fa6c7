#!/bin/bash
# A/B/n: several builds of libp3hip.so on ONE box, interleaved: tools/abn.sh <rounds> <lib1.so> <lib2.so> ... [-- bench args]
R=$1; shift
LIBS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do LIBS+=("$1"); shift; done; [ "$1" == "--" ] && shift
for i in $(seq 1 $R); do
  for L in "${LIBS[@]}"; do
    echo -n "$(basename $L): "
    P3HIP_LIB=$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timing --no-fwd "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"
  done
done
