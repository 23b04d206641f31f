#!/bin/bash
set -euo pipefail
# Board power and clocks while the headline kernel runs back to back (documentation of the clock give-back).
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"
rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -E "Power|clock|Max" | head -12
python tools/quick_bench_loop.py 6 > gpurun_out/power_loop.log 2>&1 &
PID=$!
sleep 2.5
for i in 1 2 3 4; do rocm-smi --showpower --showclocks 2>&1 | grep -E "Socket|sclk|mclk" | tr '\n' ' '; echo; sleep 0.7; done
wait $PID
tail -2 gpurun_out/power_loop.log
