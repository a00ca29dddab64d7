"""Where the host time of a steady-state decode step goes (enqueue vs fetch)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poseestimation_amd import synth
from poseestimation_amd.predict_decode import DecodeEngine
B, H, W = int(os.environ.get('PD_HC_BATCH', '64')), 368, 368
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 3
heat, paf = synth.batch(B, H, W, (1, 8), seed0=0, paf_order='xy')
dh, dp = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
engs = [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                     max_pairs_per_image=40000) for _ in range(NC)]
if len(sys.argv) > 2:      # event pair around one stage, as bench.py's timed region has it
    for e in engs:
        e.ctx.profile(stages=[sys.argv[2]])
def run(n, tq, tf):
    pend = []
    for i in range(n):
        e = engs[i % NC]
        if len(pend) == NC:
            t = time.perf_counter(); pend.pop(0).ctx.fetch(); tf[0] += time.perf_counter() - t
        t = time.perf_counter(); e.enqueue_device(dh.data_ptr(), dp.data_ptr(), B, H // 8, W // 8, (H, W)); tq[0] += time.perf_counter() - t
        pend.append(e)
    while pend:
        t = time.perf_counter(); pend.pop(0).ctx.fetch(); tf[0] += time.perf_counter() - t
run(20, [0], [0])
torch.cuda.synchronize()
tq, tf = [0.0], [0.0]
N = 400
t0 = time.perf_counter(); run(N, tq, tf); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('batch', B, 'contexts', NC, 'us/step total', dt / N * 1e6, 'enqueue', tq[0] / N * 1e6, 'fetch', tf[0] / N * 1e6)
# enqueue only (no fetch) to see the pure launch cost
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(60):
    engs[i % NC].enqueue_device(dh.data_ptr(), dp.data_ptr(), B, H // 8, W // 8, (H, W))
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print('60 enqueues: host', (t1 - t0) / 60 * 1e6, 'us each; drained after', (t2 - t0) / 60 * 1e6, 'us each')
for e in engs: e.ctx.fetch()

# GPU-side view of the same steady state: an event pair around every decode on its context's
# stream (chain = time from the first kernel's turn to the end of the last; gap = time the
# stream then sits idle until the next decode's turn)
if os.environ.get('PD_HC_EVENTS') == '1':
    streams = [torch.cuda.ExternalStream(int(e.ctx.stream)) for e in engs]
    ev = [[] for _ in engs]
    pend = []
    for i in range(400):
        c = i % NC
        e = engs[c]
        if len(pend) == NC:
            pend.pop(0).ctx.fetch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(streams[c])
        e.enqueue_device(dh.data_ptr(), dp.data_ptr(), B, H // 8, W // 8, (H, W))
        e1.record(streams[c])
        ev[c].append((e0, e1))
        pend.append(e)
    while pend:
        pend.pop(0).ctx.fetch()
    torch.cuda.synchronize()
    chain, gap = [], []
    for c in range(NC):
        for k in range(20, len(ev[c])):
            chain.append(ev[c][k][0].elapsed_time(ev[c][k][1]) * 1e3)
            gap.append(ev[c][k - 1][1].elapsed_time(ev[c][k][0]) * 1e3)
    chain.sort(); gap.sort()
    print('events: chain median %.1f us (p10 %.1f, p90 %.1f); idle gap before a chain median %.1f us (p10 %.1f, p90 %.1f)'
          % (chain[len(chain) // 2], chain[len(chain) // 10], chain[9 * len(chain) // 10],
             gap[len(gap) // 2], gap[len(gap) // 10], gap[9 * len(gap) // 10]))
