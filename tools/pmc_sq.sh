#!/bin/bash
# Shader-engine counters of every kernel of a decode (one context), to see what each kernel
# waits for.  Usage: pmc_sq.sh [bench.py workload].  Output under gpurun_out/pmc_sq.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${1:-b64_368}
OUT=$R/gpurun_out/pmc_sq
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload $WL --steps 6 --warmup 2 --contexts 1 --no-cpu"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/a -- python3 $R/bench.py $ARGS > $OUT/a.json 2> $OUT/a.log
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $OUT/b -- python3 $R/bench.py $ARGS > $OUT/b.json 2> $OUT/b.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for kind in ('a', 'b'):
    fs = glob.glob(out + '/' + kind + '/*/*counter_collection.csv')
    if not fs:
        print(kind, 'no output'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r['Kernel_Name'].split('(')[0][-28:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in agg.items():
        v = sorted(v)
        tab[k][c] = v[len(v) // 2]
cols = sorted({c for k in tab for c in tab[k]})
with open(out + '/table.txt', 'w') as fh:
    fh.write('kernel ' + ' '.join(cols) + '\n')
    for k in tab:
        fh.write(k + ' ' + ' '.join('%.3g' % tab[k].get(c, float('nan')) for c in cols) + '\n')
print(open(out + '/table.txt').read())
PY
