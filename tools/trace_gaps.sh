#!/bin/bash
# Kernel trace of a one-context bench run: start/end stamps of every kernel, for looking
# at the gaps between the kernels of one decode.  Output under gpurun_out/gaps.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/gaps
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu --steps 40 --warmup 6 --contexts 1 > $OUT/bench.json 2> $OUT/trace.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
f = glob.glob(out + '/trace/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
rows = rows[-14 * 30:]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev = None
for r in rows:
    k = r['Kernel_Name'].split('(')[0][:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[k].append(e - s)
    if prev is not None: gap[k].append(s - prev)
    prev = e
with open(out + '/gaps.txt', 'w') as fh:
    for k in dur:
        d = sorted(dur[k]); g = sorted(gap[k]) or [0]
        fh.write('%-42s n=%3d dur_med=%7.2f us gap_before_med=%7.2f us\n' % (k, len(d), d[len(d)//2] / 1e3, g[len(g)//2] / 1e3))
print(open(out + '/gaps.txt').read())
PY
