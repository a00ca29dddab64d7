#!/bin/bash
# Kernel statistics and memory counters of the four-scale resize + average path.
# Output under gpurun_out/ms.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/ms
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/bench_multiscale.py 64 > $OUT/trace.txt 2> $OUT/trace.log
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/fetch -- python3 $R/tools/bench_multiscale.py 64 > $OUT/fetch.txt 2> $OUT/fetch.log
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE TCC_MISS_sum --output-format csv -d $OUT/write -- python3 $R/tools/bench_multiscale.py 64 > $OUT/write.txt 2> $OUT/write.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/*/*kernel_stats.csv')[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print('%-50s calls %5s avg_us %10.1f total_ms %8.2f' % (r['Name'][:50], r['Calls'], float(r['AverageNs']) / 1e3, float(r['TotalDurationNs']) / 1e6))
for kind in ('fetch', 'write'):
    fs = glob.glob(out + '/' + kind + '/*/*counter_collection.csv')
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'resize' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0], r['Counter_Name'], r['Grid_Size'])].append(float(r['Counter_Value']))
    for k, v in sorted(agg.items()):
        print(kind, k, 'mean', sum(v) / len(v), 'n', len(v))
PY
