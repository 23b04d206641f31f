#!/bin/bash
set -euo pipefail
# Collect PMC counters for the PAL-D bench in separate passes (gpurun refuses --pmc with trace domains).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"      # the repository root, wherever the script is started from
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc
mkdir -p $OUT
N=${1:-1000}
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/p1 -- python tools/quick_bench.py $N > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python tools/quick_bench.py $N > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python tools/quick_bench.py $N > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python tools/quick_bench.py $N > $OUT/p4.log 2>&1
python tools/pmc_summary.py ${2:-}
# (tools/pmc_summary.py prints the per-kernel counter means of these passes)
