#!/bin/bash
# Development build of the headline instance only: tools/dev_build.sh NAME [extra hipcc flags] -> build_ab/libNAME.so (+ ISA)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
# ablation switches (CM_EXP_*, CM_DEV_ROLE) compile only under -DCM_EXPERIMENTS, which cm_plan_describe reports
EXPERIMENTS=""
case " $* " in *" -DCM_EXP_"*|*" -DCM_DEV_ROLE"*) EXPERIMENTS="-DCM_EXPERIMENTS";; esac
mkdir -p $ROOT/build_ab /tmp/cm_build/$NAME
cd /tmp/cm_build/$NAME   # temporaries (ISA: /tmp/cm_build/NAME/*.s) stay out of the tree
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fno-slp-vectorize -DCM_DEV_PALD_ONLY $EXPERIMENTS "$@" -save-temps \
  -Rpass-analysis=kernel-resource-usage -o $ROOT/build_ab/lib$NAME.so $ROOT/color_modem_amd/csrc/cm_api.hip 2>&1 \
  | grep -E "error|demod_(pair_)?kernel" -A5 | grep -E "error|VGPRs:|SGPRs:|Occupancy|LDS Size|ScratchSize" || true
echo "scratch instructions: $(grep -c scratch_ cm_api-hip-amdgcn-amd-amdhsa-gfx950.s)"
