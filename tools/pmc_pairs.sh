#!/bin/bash
# Counter passes over a few steps of the 656 x 656 / 30 people workload (one context), for
# the pair-scoring kernel.  Output under gpurun_out/pmc_pairs.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/pmc_pairs
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--workload b256_656 --steps 6 --warmup 2 --contexts 1 --no-cpu"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/sq -- python3 $R/bench.py $ARGS > $OUT/sq.json 2> $OUT/sq.log
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- python3 $R/bench.py $ARGS > $OUT/tcp.json 2> $OUT/tcp.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d $OUT/tcc -- python3 $R/bench.py $ARGS > $OUT/tcc.json 2> $OUT/tcc.log
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum --output-format csv -d $OUT/ta -- python3 $R/bench.py $ARGS > $OUT/ta.json 2> $OUT/ta.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for kind in ('sq', 'tcp', 'tcc', 'ta'):
    fs = glob.glob(out + '/' + kind + '/*/*counter_collection.csv')
    if not fs:
        print(kind, 'no output'); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if 'k_pair_scores' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        v = sorted(v)
        print(kind, k, 'median', v[len(v) // 2], 'n', len(v))
PY
