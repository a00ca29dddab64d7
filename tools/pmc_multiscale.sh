#!/bin/bash
# Shader, LDS and memory counters of the kernels of the multi-scale decode (bench.py --workload
# ms4_b128_368, a few steps, one context), each counter group in a run of its own.  Output under
# gpurun_out/pmc_ms; the table at the end is what goes to profiles/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_ms
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="$R/bench.py --workload ms4_b128_368 --steps 3 --warmup 1 --contexts 1 --no-cpu --no-extras"
timeout -s KILL 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.txt 2> $OUT/trace.log || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d $OUT/a -- python3 $ARGS > $OUT/a.txt 2> $OUT/a.log || exit 1
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- python3 $ARGS > $OUT/b.txt 2> $OUT/b.log || exit 1
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ARGS > $OUT/f.txt 2> $OUT/f.log || exit 1
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ARGS > $OUT/w.txt 2> $OUT/w.log || exit 1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for kind in ('a', 'b', 'fetch', 'write'):
    fs = glob.glob(out + '/' + kind + '/*/*counter_collection.csv')
    if not fs:
        print(kind, 'no output'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r['Kernel_Name'].lstrip('void ').startswith('k_'):
            agg[(r['Kernel_Name'].split('(')[0] + ' grid ' + r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in agg.items():
        tab[k][c] = sum(v) / len(v)
with open(out + '/summary.txt', 'w') as fh:
    for k in sorted(tab):
        print(k, file=fh)
        for c in sorted(tab[k]):
            print('    %-28s %.5g' % (c, tab[k][c]), file=fh)
    for f in glob.glob(out + '/trace/*/*kernel_stats.csv'):
        print(open(f).read(), file=fh)
print(open(out + '/summary.txt').read())
PY
