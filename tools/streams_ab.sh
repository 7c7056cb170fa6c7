#!/bin/bash
# Same-box A/B of the side-stream switches (ops.SIDE): tools/streams_ab.sh [rounds]  ->  ms per captured train step per configuration,
# interleaved over the rounds (the pool's boxes differ by ~3 %, so only numbers of ONE invocation compare).  VERDICT r03 item 3.
R=${1:-3}
CONFIGS=("base:" "sn:P3_SIDE_SN=1" "dw:P3_SIDE_DW=1" "stem:P3_SIDE_STEM=1" "sn+stem:P3_SIDE_SN=1 P3_SIDE_STEM=1" "all:P3_SIDE_SN=1 P3_SIDE_DW=1 P3_SIDE_STEM=1")
for i in $(seq 1 $R); do
  for c in "${CONFIGS[@]}"; do
    name=${c%%:*}; envs=${c#*:}
    ms=$(env $envs python bench.py --lean --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['final_loss'])" 2>/dev/null || echo "FAILED")
    echo "round $i  $name  $ms"
  done
done
