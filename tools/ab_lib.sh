#!/bin/bash
# A/B of two builds of the library on the bench workloads: tools/ab_lib.sh <other.so> <workload>...
# Alternates the in-tree build and the other one, three runs each; prints images/s.
other=$1; shift
mkdir -p gpurun_out
for w in "$@"; do
  for i in 1 2 3; do
    for which in tree other; do
      if [ $which = other ]; then export PD_LIBRARY=$other; else unset PD_LIBRARY; fi
      timeout -k 10 200 python bench.py --workload $w --no-cpu --no-stage-timing > gpurun_out/ab.json 2> gpurun_out/ab.err || exit 1
      python -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print('$w','$which',round(d['value']))"
    done
  done
done
