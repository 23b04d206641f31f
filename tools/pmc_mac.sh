#!/bin/bash
set -euo pipefail
# rocprofv3 kernel stats and FETCH / WRITE counters of the MAC kernels: tools/pmc_mac.sh <tag>  (on the GPU box)
ROOT="$(cd "$(dirname "$0")/.." && pwd)"      # the repository root, wherever the script is started from
cd /tmp && export TMPDIR=/tmp && cd "$ROOT"
OUT=gpurun_out/mac_${1:?tag}
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python tools/quick_bench_mac.py 1000 > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/p1 -- python tools/quick_bench_mac.py 1000 > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/p2 -- python tools/quick_bench_mac.py 1000 > $OUT/p2.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for p in glob.glob(out + '/stats/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'mac_' in r['Name']: print('%-40s calls %s avg %.3f ms min %.3f ms' % (r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e6, float(r['MinNs']) / 1e6))
tot = collections.defaultdict(list)
for p in glob.glob(out + '/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'mac_' in r['Kernel_Name']: tot[(r['Kernel_Name'][:32], r['Counter_Name'])].append(float(r['Counter_Value']))
for k in sorted(tot): print('%-34s %-11s mean %.3f GB raw (n=%d)' % (k[0], k[1], sum(tot[k]) / len(tot[k]) * 1024 / 1e9, len(tot[k])))
print('algorithmic per launch (1000 frames x 576 rows x 12960 B): 7.465 GB = 2.488 GB composite + 4.977 GB rgb')
PY
