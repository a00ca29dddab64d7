#!/bin/bash
# Profiles of one bench workload on the GPU box: per-kernel times (kernel trace) and, in
# passes of their own, the memory-side byte counters.  Usage: profile.sh [workload]
# Output under gpurun_out/profile_<workload>; tools/summarize_profile.py copies the
# summaries to profiles/.
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${1:-b64_368}
OUT=$R/gpurun_out/profile_$WL
# PROFILE_NO_PMC=1: only the two bench runs (after summarize_profile.py has refreshed
# profiles/pmc_<workload>.json, so that the bench lines carry the traffic of this build)
if [ -z "$PROFILE_NO_PMC" ]; then rm -rf $OUT; fi
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SHORT="--workload $WL --no-cpu --no-extras --steps 6 --warmup 2 --contexts 1"
timeout -k 10 300 python3 $R/bench.py --workload $WL > $OUT/bench.json 2> $OUT/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py --workload $WL --no-cpu --no-extras > $OUT/trace_bench.json 2> $OUT/trace.log
if [ -n "$PROFILE_NO_PMC" ]; then echo done $WL; exit 0; fi
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $R/bench.py $SHORT > $OUT/fetch_bench.json 2> $OUT/fetch.log
timeout -s KILL 150 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $R/bench.py $SHORT > $OUT/write_bench.json 2> $OUT/write.log
echo done $WL
