"""Throughput of the fetch pipeline (four contexts, replayed graphs) in windows of 100 steps
over a long run: does the rate hold, and if not, when does it change?"""
import os, sys, time
os.environ['GPU_MAX_HW_QUEUES'] = os.environ.get('PD_QUEUES', '8')
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from poseestimation_amd import synth
from poseestimation_amd.predict_decode import DecodeEngine
B, H, W = 64, 368, 368
NC = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
heat, paf = synth.batch(B, H, W, (1, 8), seed0=0, paf_order='xy')
dh, dp = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
engs = [DecodeEngine(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                     max_pairs_per_image=40000) for _ in range(NC)]
if len(sys.argv) > 3:
    for e in engs:
        e.ctx.profile(stages=[sys.argv[3]])
if os.environ.get('PD_GC') == 'off':
    import gc
    gc.collect(); gc.freeze()
pend = []
THREADS = os.environ.get('PD_SUBMIT') == '1'   # launches issued by the contexts' launcher threads
def fetch(e):
    if THREADS:
        e.ctx.flush()
    return e.ctx.fetch()
def step(i):
    e = engs[i % NC]
    if len(pend) == NC:
        fetch(pend.pop(0))
    if THREADS:
        e.ctx.submit(dh.data_ptr(), dp.data_ptr(), B, H // 8, W // 8, H, W, 0.1, 0.05)
    else:
        e.enqueue_device(dh.data_ptr(), dp.data_ptr(), B, H // 8, W // 8, (H, W))
    pend.append(e)
for i in range(5 * NC):
    step(i)
torch.cuda.synchronize()
marks = [time.perf_counter()]
for i in range(N):
    step(i)
    if (i + 1) % 100 == 0:
        marks.append(time.perf_counter())
while pend:
    fetch(pend.pop(0))
us = [(b - a) / 100 * 1e6 for a, b in zip(marks, marks[1:])]
print('us per step by window of 100:', ' '.join('%.0f' % u for u in us))
print('mean', sum(us) / len(us), 'min', min(us), 'max', max(us))
