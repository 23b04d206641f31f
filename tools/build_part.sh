#!/bin/bash
# tools/build_part.sh PART [extra flags]: compile one CM_PART of cm_api.hip into color_modem_amd/_build (ISA + resource log in /tmp/cm_build/pPART), then link the library
# NOLINK=1 skips the link (several parts side by side: link with the last one)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PART=$1; shift   # 1 .. 7
mkdir -p /tmp/cm_build/p$PART $ROOT/color_modem_amd/_build
cd /tmp/cm_build/p$PART
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DCM_PART=$PART "$@" -save-temps -Rpass-analysis=kernel-resource-usage -c \
  -o $ROOT/color_modem_amd/_build/cm_api_part$PART.o $ROOT/color_modem_amd/csrc/cm_api.hip > build.log 2>&1 || { grep -E "error" -A3 build.log | head -40; exit 1; }
if [ -z "$NOLINK" ]; then
  OBJS=$(ls $ROOT/color_modem_amd/_build/cm_api_part*.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $ROOT/color_modem_amd/libcolor_modem_hip.so $OBJS
fi
echo "part $PART built; log /tmp/cm_build/p$PART/build.log"
