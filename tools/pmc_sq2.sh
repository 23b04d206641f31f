#!/bin/bash
set -euo pipefail
# SQ counter passes on tools/quick_bench.py; usage: tools/pmc_sq2.sh TAG [CM_LIB path]
TAG=${1:?tag}
if [ -n "${2:-}" ]; then export CM_LIB=$2; fi
ROOT="$(cd "$(dirname "$0")/.." && pwd)"      # the repository root, wherever the script is started from
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/p1 -- python tools/quick_bench.py 1000 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python tools/quick_bench.py 1000 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python tools/quick_bench.py 1000 > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH --output-format csv -d $OUT/p4 -- python tools/quick_bench.py 1000 > $OUT/p4.log 2>&1
python - $OUT <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
tot = collections.defaultdict(list)
for p in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'demod_' in r['Kernel_Name']: tot[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in tot.items()}
for c in sorted(m): print('   %-24s %.4g' % (c, m[c]))
wc = m['SQ_WAVE_CYCLES']
g = lambda k: m.get(k, float('nan'))
print('fractions of WAVE_CYCLES: active_any %.3f wait_any %.3f wait_inst_any %.3f active_valu %.3f' % (g('SQ_ACTIVE_INST_ANY')/wc, g('SQ_WAIT_ANY')/wc, g('SQ_WAIT_INST_ANY')/wc, g('SQ_ACTIVE_INST_VALU')/wc))
print('per wave: valu %.0f salu %.0f smem %.0f lds %.0f vmem %.0f' % tuple(g(k)/m['SQ_WAVES'] for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_SMEM','SQ_INSTS_LDS','SQ_INSTS_VMEM')))
PY
for f in $OUT/p*.log; do tail -1 $f | cut -c1-80; done
