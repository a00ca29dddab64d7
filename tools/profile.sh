#!/bin/bash
# Profiles of the default bench command on the GPU box: per-kernel times (kernel trace)
# and, in passes of their own, the memory-side byte counters.  Output under gpurun_out/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/profile
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python3 $R/bench.py > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --no-cpu > $OUT/trace_bench.json 2> $OUT/trace.log
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py --no-cpu --steps 12 --warmup 3 --contexts 1 > $OUT/fetch_bench.json 2> $OUT/fetch.log
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py --no-cpu --steps 12 --warmup 3 --contexts 1 > $OUT/write_bench.json 2> $OUT/write.log
echo done
