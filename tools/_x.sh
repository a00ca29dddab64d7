run() { timeout -k 10 200 python3 bench.py --no-cpu --no-extras --no-stage-timing "$@" > gpurun_out/ab.json 2> gpurun_out/ab.err || { tail -3 gpurun_out/ab.err; return; }
  python3 -c "import json;d=json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]);print('$LABEL',round(d['value']), round(d['ms_per_step']*1000,2))"; }
export PD_BENCH_FORCE_DIST=1
for c in 3 4; do for g in 1 2 3 4; do PD_BENCH_GATHER_DEPTH=$g LABEL="RCCL one rank ctx $c depth $g" run --contexts $c; done; done
for q in 4 5 6; do LABEL="RCCL one rank ctx 3 queues $q" run --hw-queues $q; done
