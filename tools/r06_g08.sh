#!/bin/bash
# r06 g08: A-stationary kernel v5 (16x16x32 MFMA, four independent accumulator chains): check, timing against the tile kernels, SQ counters
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/mb_as_8.txt
: > $O
timeout 300 python tools/mb_as.py check >> $O 2>&1
timeout 300 python tools/mb_as.py time >> $O 2>&1
P3_AS_VAR=2 timeout 200 python tools/mb_as.py as >> $O 2>&1
for v in 0; do
  for row in "qkv " "fc1      1536x384  GELU+aux" "dX fc2"; do
    rm -rf /tmp/pq
    P3_AS_VAR=$v timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/pq -o q -- python tools/mb_as.py one "$row" > /tmp/pq.log 2>&1
    echo "== P3_AS_VAR=$v  $row" >> $O
    python tools/pmc_kernels.py /tmp/pq gemm_x3_as >> $O 2>&1
  done
done
grep -v amdgpu.ids $O | tail -75
