#!/bin/bash
# Sweep of one environment variable over a bench workload: tools/sweep_env.sh <VAR> <workload> <value>...
var=$1; w=$2; shift 2
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    export $var=$v
    timeout -k 10 200 python bench.py --workload $w --no-cpu --no-stage-timing > gpurun_out/sw.json 2> gpurun_out/sw.err || exit 1
    python -c "import json;d=json.loads(open('gpurun_out/sw.json').read().strip().splitlines()[-1]);print('$w','$var=$v',round(d['value']))"
  done
done
