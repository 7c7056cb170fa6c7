#!/bin/bash
# r06 g31: MFMA / VALU co-issue probe
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue tools/probe/coissue_probe.hip 2>/dev/null
timeout 120 /tmp/coissue > gpurun_out/r06_coissue_probe.txt 2>&1
cat gpurun_out/r06_coissue_probe.txt
