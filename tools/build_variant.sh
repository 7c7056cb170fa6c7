#!/bin/bash
# Link a variant of libp3hip.so for a same-box A/B: tools/build_variant.sh <out.so> <file.hip> [extra hipcc flags...]
# The named source is compiled from the working tree (with the extra flags, e.g. -DP3_X=1) and linked with the cached objects of the rest.
set -e
OUT=$1; SRC=$2; shift 2
C=pixelspointspolygons_amd/csrc
python -c "from pixelspointspolygons_amd.build import build_library; build_library(verbose=False)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $C/$SRC -o /tmp/variant_$SRC.o
OBJS=$(ls $C/_obj/*.o | grep -v "/$SRC.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT $OBJS /tmp/variant_$SRC.o
echo "built $OUT"
