#!/bin/bash
# Rehearsal of bench.py's N > 1 code path on a box with one GPU: N ranks under torch.distributed.run, all on the
# GPUs there are, gathers over gloo (RCCL refuses two ranks on one GPU).  Checks that every workload's run ends with
# rank 0's JSON line; the rates mean nothing.  Usage: tools/rehearse_ranks.sh [N]   (output under gpurun_out/)
N=${1:-2}
mkdir -p gpurun_out
export PD_BENCH_BACKEND=gloo
port=29550
for w in ${WORKLOADS:-b64_368 stream_b1024_368 ms4_b128_368 b256_656}; do
  port=$((port + 1))
  timeout -k 10 280 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $port \
      bench.py --gpus $N --workload $w --steps 12 --warmup 3 > gpurun_out/rehearse_${w}_$N.json 2> gpurun_out/rehearse_${w}_$N.err
  rc=$?
  echo "$w ranks $N rc=$rc: $(tail -1 gpurun_out/rehearse_${w}_$N.json | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d[\"n_gpus\"], round(d[\"value\"]), d.get(\"parity_ok\"), d.get(\"parity_checked_images\"), d.get(\"persons_last_step\"))")"
  if [ $rc -ne 0 ]; then grep -v "socket.cpp\|amdgpu.ids" gpurun_out/rehearse_${w}_$N.err | tail -15; exit 1; fi
done
