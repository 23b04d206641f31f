#!/bin/bash
# A/B of library builds on any bench command: tools/ab_any.sh "python tools/quick_bench_mod.py secam" libA.so libB.so ...
set -euo pipefail
cd "$(cd "$(dirname "$0")/.." && pwd)"
CMD="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "; CM_LIB=$PWD/$lib $CMD 2>/dev/null | tail -1
  done
done
