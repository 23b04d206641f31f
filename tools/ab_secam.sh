#!/bin/bash
# A/B of SECAM decoder builds: tools/ab_secam.sh libA.so libB.so ...  (alternating, three rounds)
set -euo pipefail
cd "$(cd "$(dirname "$0")/.." && pwd)"
for round in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib: "; CM_LIB=$PWD/$lib python tools/quick_bench_secam.py 1000 | tail -1
  done
done
