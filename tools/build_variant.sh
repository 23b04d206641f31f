#!/bin/bash
# tools/build_variant.sh NAME PART [flags]: one CM_PART of cm_api.hip rebuilt with extra flags and linked with the other parts' objects of
# color_modem_amd/_build into build_ab/libNAME.so (A/B builds: CM_LIB=build_ab/libNAME.so picks it); ISA + resource log in /tmp/cm_build/NAME
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; PART=$2; shift; shift
mkdir -p /tmp/cm_build/$NAME $ROOT/build_ab
cd /tmp/cm_build/$NAME
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM_PART=$PART "$@" -save-temps -Rpass-analysis=kernel-resource-usage -c \
  -o /tmp/cm_build/$NAME/part$PART.o $ROOT/color_modem_amd/csrc/cm_api.hip > build.log 2>&1 || { grep -E "error" -A3 build.log | head -40; exit 1; }
OBJS=$(ls $ROOT/color_modem_amd/_build/cm_api_part*.o | grep -v "part$PART.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/build_ab/lib$NAME.so $OBJS /tmp/cm_build/$NAME/part$PART.o
echo "variant $NAME built (part $PART $*)"
