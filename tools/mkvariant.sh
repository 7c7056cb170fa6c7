#!/bin/bash
# A variant build of libp3hip.so for a same-box A/B (tools/ab.sh, P3HIP_LIB): tools/mkvariant.sh <out.so> <csrc file> [extra hipcc flags...]
# recompiles ONE source with the extra flags (e.g. -DP3_PX_PRIO=0) and links it with the up-to-date objects of the regular build
set -e
OUT=$1; SRC=$2; shift 2
D=pixelspointspolygons_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $D/$SRC -o /tmp/variant_$SRC.o
OBJS=$(ls $D/_obj/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS /tmp/variant_$SRC.o
echo "built $OUT"
