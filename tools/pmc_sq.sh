#!/bin/bash
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"      # the repository root, wherever the script is started from
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_sq
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM --output-format csv -d $OUT/p1 -- python tools/quick_bench.py 1000 > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --output-format csv -d $OUT/p2 -- python tools/quick_bench.py 1000 > $OUT/p2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python tools/quick_bench.py 1000 > $OUT/p3.log 2>&1
python - <<'PY'
import csv, glob, collections
tot = collections.defaultdict(list)
for p in glob.glob('gpurun_out/pmc_sq/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'demod_' in r['Kernel_Name']: tot[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in tot.items()}
for c in sorted(m): print('   %-24s %.4g' % (c, m[c]))
wc = m['SQ_WAVE_CYCLES']
print('fractions of WAVE_CYCLES: active_any %.3f wait_any %.3f wait_inst_any %.3f active_valu %.3f' % (m['SQ_ACTIVE_INST_ANY']/wc, m['SQ_WAIT_ANY']/wc, m['SQ_WAIT_INST_ANY']/wc, m['SQ_ACTIVE_INST_VALU']/wc))
print('per wave: valu %.0f salu %.0f smem %.0f lds %.0f vmem %.0f' % tuple(m[k]/m['SQ_WAVES'] for k in ('SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_SMEM','SQ_INSTS_LDS','SQ_INSTS_VMEM')))
PY
tail -1 $OUT/p1.log | cut -c1-60
