#!/bin/bash
# Shader and cache counters of the multi-scale kernels.  Output under gpurun_out/pmc_ms.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_ms
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P="python3 $R/tools/bench_multiscale.py 32"
timeout -s KILL 150 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU --output-format csv -d $OUT/a -- $P > $OUT/a.txt 2> $OUT/a.log
timeout -s KILL 150 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/b -- $P > $OUT/b.txt 2> $OUT/b.log
timeout -s KILL 150 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum --output-format csv -d $OUT/c -- $P > $OUT/c.txt 2> $OUT/c.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tab = collections.defaultdict(dict)
for kind in ('a', 'b', 'c'):
    fs = glob.glob(out + '/' + kind + '/*/*counter_collection.csv')
    if not fs:
        print(kind, 'no output'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'resize' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0] + ':' + r['Grid_Size'], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in agg.items():
        tab[k][c] = sum(v) / len(v)
for k in sorted(tab):
    print(k)
    for c in sorted(tab[k]):
        print('    %-40s %.4g' % (c, tab[k][c]))
PY
