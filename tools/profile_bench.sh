#!/bin/bash
# Round profiles of the headline benchmark ON THE DRIVER'S OWN COMMAND (python3 bench.py --gpus 1 --steps 20 --warmup 5):
#   kernel trace + stats, then PMC passes (memory traffic, SQ issue / wait / instruction counters), each in its own run.
# usage (on the GPU box, from the repository root):  tools/profile_bench.sh <tag>     -> gpurun_out/prof_<tag>/summary.txt
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
TAG="${1:?tag}"
OUT="$ROOT/gpurun_out/prof_$TAG"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
# the headline passes time the PAL-D kernel alone (--other-configs 0: BASELINE configs 3 / 4 have their own run below)
ARGS="bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --other-configs 0"
python3 $ARGS > "$OUT/bench_line.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -- python3 $ARGS > "$OUT/kt.log" 2>&1
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/m1" -- python3 $ARGS > "$OUT/m1.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/m2" -- python3 $ARGS > "$OUT/m2.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d "$OUT/s1" -- python3 $ARGS > "$OUT/s1.log" 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d "$OUT/s2" -- python3 $ARGS > "$OUT/s2.log" 2>&1
# a sustained run beside the 20-step burst: 2000 steps = about 5 s of back-to-back launches
python3 bench.py --gpus 1 --steps 2000 --warmup 5 --cpu-sample 0 --other-configs 0 > "$OUT/bench_line_sustained.json" 2>> "$OUT/bench.err"
# the driver's default line (other_configs: NTSC 3D comb, SECAM encode / decode / round trip; cpu_baseline) and its kernel trace
python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench_line_default_run.json" 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_all" -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > "$OUT/kt_all.log" 2>&1
python3 tools/profile_summary.py "$OUT" > "$OUT/summary.txt"
python3 - "$OUT" >> "$OUT/summary.txt" <<'PY'
import json, sys
for name in ('bench_line.json', 'bench_line_sustained.json', 'bench_line_default_run.json'):
    try:
        d = json.loads([l for l in open(sys.argv[1] + '/' + name) if l.startswith('{')][-1])
        print('%-28s steps %5d  ms_per_step %.4f  kernel_ms %.4f  %.0f Mpixels/s  frac %.4f' % (name, d['steps'], d['ms_per_step'], d['roofline']['kernel_ms'], d['value'], d['roofline']['frac']))
        for c in d.get('other_configs', []):
            print('   other_configs: %-96s %.4f ms  %.0f Mpixels/s  frac %.4f  check %.2g' % (c['workload'][:96], c['ms'], c['mpixels_s'], c['roofline']['frac'], c['check']['max_rel_err']))
        if d.get('other_configs_error'):
            print('   other_configs FAILED: %s' % d['other_configs_error'])
        print('   traffic_source: %s' % d['roofline'].get('traffic_source'))
        print('   sclk during the timed steps: %s; VALU-limited %.4f ms at 2.4 GHz against %.4f measured' % (d['roofline_valu'].get('sclk_MHz_during_timed_steps'), d['roofline_valu'].get('valu_limited_ms_at_2400MHz', float('nan')), d['roofline']['kernel_ms']))
    except Exception as e:
        print(name, 'missing', e)
PY
# the line as the driver will print it once this run's traffic.json is committed: same command, traffic from the passes above
cp "$OUT/traffic.json" "$ROOT/profiles/traffic.json"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --other-configs 0 > "$OUT/bench_line_with_traffic.json" 2>> "$OUT/bench.err"
cat "$OUT/summary.txt"
