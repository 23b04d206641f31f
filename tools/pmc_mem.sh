#!/bin/bash
set -euo pipefail
# usage: pmc_mem.sh <tag>  (CM_LIB may point at an experimental build)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"      # the repository root, wherever the script is started from
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/pmc_${1:?tag}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- python tools/quick_bench.py 1000 > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/p4 -- python tools/quick_bench.py 1000 > $OUT/p4.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
tot = collections.defaultdict(list)
for p in glob.glob(sys.argv[1] + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'demod_' in r['Kernel_Name']: tot[r['Counter_Name']].append(float(r['Counter_Value']))
m = {c: sum(v) / len(v) for c, v in tot.items()}
print(sys.argv[1], 'FETCH raw GB %.3f  WRITE GB %.3f  L2 hit %.3f' % (m['FETCH_SIZE'] * 1024 / 1e9, m['WRITE_SIZE'] * 1024 / 1e9, m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])))
PY
tail -1 $OUT/p3.log | cut -c1-60
