#!/bin/bash
# SQ / memory counters of any bench command: tools/pmc_any.sh TAG KERNEL_SUBSTRING -- python3 tools/... ARGS   -> gpurun_out/pmc_TAG/summary.txt
# (program right after --: no env / bash -c hops under rocprofv3; each counter group in its own run)
set -euo pipefail
TAG=$1; KSUB=$2; shift 3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$ROOT/gpurun_out/pmc_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- "$@" > "$OUT/kt.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/s1" -- "$@" > "$OUT/s1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/s2" -- "$@" > "$OUT/s2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/m1" -- "$@" > "$OUT/m1.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/m2" -- "$@" > "$OUT/m2.log" 2>&1
python3 tools/pmc_any_summary.py "$OUT" "$KSUB" > "$OUT/summary.txt"
cat "$OUT/summary.txt"
