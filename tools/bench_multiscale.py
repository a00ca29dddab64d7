"""Four-scale configuration (predict.py:49-72 with scale_search 0.5, 1, 1.5, 2 on 368x368
images): stage times of resize + average + decode from the averaged maps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from poseestimation_amd import synth
from poseestimation_amd.predict_decode import DecodeEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = W = 368
scales = [0.5, 1.0, 1.5, 2.0]
base = [[synth.scene(H, W, 1 + b % 8, 500 + b, paf_order='xy', scale=s) for b in range(8)] for s in scales]
heats = [np.stack([base[m][b % 8][0] for b in range(B)]) for m in range(4)]
pafs = [np.stack([base[m][b % 8][1] for b in range(B)]) for m in range(4)]
pads = [(0, 0)] * 4
eng = DecodeEngine(B, H, W, mode='D', max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
eng.decode_multiscale(heats, pafs, pads, (H, W))
eng.ctx.profile(True)
eng.ctx.profile_read()
N = 3
t0 = time.perf_counter()
for _ in range(N):
    res = eng.decode_multiscale(heats, pafs, pads, (H, W))
dt = (time.perf_counter() - t0) / N
prof = eng.ctx.profile_read()
print('B', B, 'ms per batch (host arrays in, results out)', dt * 1e3, 'images/s', B / dt)
print({k: (round(v[0] / N, 3), v[1] // N) for k, v in prof.items() if v[1]})
n, Nf = sum(h.shape[1] * h.shape[2] for h in heats), H * W
Ns = sum(64 * h.shape[1] * h.shape[2] for h in heats)
alg = B * (228.0 * n + 2 * 228.0 * Ns + 57.0 * Nf * 8)
print('resize stage: algorithmic GB', alg / 1e9, 'GB/s', alg / 1e9 / (prof['resize'][0] / N / 1e3))
