"""Parity soak: many random synthetic images decoded on the GPU (through the C ABI, batched path) and
every one of them compared with the CPU restatement in oracle/decode_ref.py -- peaks and
connections bit for bit in position and membership, scores to 1e-5 relative, person rows.

The oracle runs in worker processes (numpy / scipy only, they never touch the GPU); the decode
runs in this process.  Geometries, thresholds, people counts and noise levels vary from batch to
batch; both reference variants (mode P = utils/postprocess.py, mode D = predict.detect) are
covered.  --multiscale: the test-time scale loop of predict.py:49-72 instead (network outputs of
two to four scales, x8 bicubic, resize, average, then the decode of the float64 averages).
Usage: parity_soak.py [--minutes M] [--workers N] [--seed S] [--multiscale]; prints one summary
line per batch and a total, exits non-zero at the first mismatch.
"""
import argparse
import multiprocessing as mp
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

RTOL = 1e-5


def oracle_image(job):
    """Worker: (heat, paf, mode, H, W, t1, t2) -> (peaks, conns, special, subset)."""
    from oracle import decode_ref as ref
    heat, paf, mode, H, W, t1, t2 = job
    hf = ref.resize_cubic(heat, fx=8, fy=8)[:H, :W].astype(np.float64)
    pf = ref.resize_cubic(paf, fx=8, fy=8)[:H, :W].astype(np.float64)
    peaks = ref.get_peaks(hf, t1, mode, use_scipy=True)
    conns, special = ref.get_connections(peaks, pf, t2, mode)
    subset = ref.multi_person_parse(peaks, conns, special, mode)
    return peaks, conns, special, subset


def oracle_multiscale(job):
    """Worker: (heats, pafs, pads, H, W, mode) -> (candidate, peaks, conns, special, subset)."""
    from oracle import decode_ref as ref
    heats, pafs, pads, H, W, mode = job
    heat_avg, paf_avg = ref.upsample_average(heats, pafs, pads, (H, W))
    cand, peaks, subset = ref.decode(heat_avg, paf_avg, 0.1, 0.05, mode)
    conns, special = ref.get_connections(peaks, paf_avg, 0.05, mode)
    return cand, peaks, conns, special, subset


MS_GEOMETRIES = [  # (H, W, batch, scales)
    (368, 368, 16, (0.5, 1.0, 1.5, 2.0)),
    (184, 248, 16, (0.5, 1.0, 1.5, 2.0)),
    (200, 312, 12, (1.0, 2.0)),
    (368, 368, 16, (0.5, 1.0, 1.5)),
    (120, 152, 16, (1.0, 1.5, 2.0)),
    (361, 275, 12, (0.5, 1.0, 1.5, 2.0)),   # no multiple of 8: pads in every scale
    (200, 560, 8, (0.5, 1.0, 1.5)),         # more than 64 cell columns: the general form of the bound kernel
    (480, 640, 4, (0.38, 0.77, 1.15, 1.53)),  # a COCO-sized image at predict.py:49's scales: beyond the sparse stage's staging areas
]


def soak_multiscale(args, pool):
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    from tests import golden_io as gio
    t_end = time.time() + 60.0 * args.minutes
    seed, rounds = args.seed, 0
    tot = dict(images=0, peaks=0, conns=0, persons=0)
    while time.time() < t_end:
        for (H, W, B, scales) in MS_GEOMETRIES:
            if time.time() >= t_end:
                break
            mode = 'D' if rounds % 2 == 0 else 'P'
            heats, pafs = [], []
            for sc in scales:
                outs = [synth.scene(H, W, (seed + b) % 7, seed + b, paf_order='xy' if mode == 'D' else 'yx', scale=sc)
                        for b in range(B)]
                heats.append(np.stack([o[0] for o in outs]))
                pafs.append(np.stack([o[1] for o in outs]))
            pads = [(h.shape[1] * 8 - int(round(H * sc)), h.shape[2] * 8 - int(round(W * sc))) for h, sc in zip(heats, scales)]
            seed += B
            eng = DecodeEngine(B, H, W, mode=mode, max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
            res = eng.decode_multiscale(heats, pafs, pads, (H, W))
            want = pool.map(oracle_multiscale, [([h[b] for h in heats], [p[b] for p in pafs], pads, H, W, mode)
                                                for b in range(B)])
            n = dict(peaks=0, conns=0, persons=0)
            for b in range(B):
                cand, peaks, conns, special, subset = want[b]
                try:
                    gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
                    np.testing.assert_array_equal(res.candidate(b), cand)
                    g_conns, g_special = res.connections(b)
                    assert g_special == special
                    gio.assert_conns_equal(g_conns, conns, RTOL)
                    gio.assert_persons_equal(res.subset(b), subset, RTOL)
                except AssertionError as e:
                    print('MISMATCH: multi-scale %dx%d scales %s mode %s seed %d image %d: %s'
                          % (H, W, scales, mode, seed - B, b, e), flush=True)
                    sys.exit(1)
                n['peaks'] += sum(len(p) for p in peaks)
                n['conns'] += sum(len(c) for c in conns)
                n['persons'] += len(subset)
            eng.close()
            tot['images'] += B
            for k in n:
                tot[k] += n[k]
            print('%4dx%-4d scales %-20s mode %s: %3d images equal (%5d peaks, %5d connections, %4d persons)'
                  % (H, W, scales, mode, B, n['peaks'], n['conns'], n['persons']), flush=True)
        rounds += 1
    print('TOTAL (multi-scale): %(images)d images, %(peaks)d peaks, %(conns)d connections, %(persons)d persons compared '
          'with the oracle: all equal' % tot)


GEOMETRIES = [  # (H, W, batch, people range, max_pairs)
    (368, 368, 64, (0, 10), 40000),
    (200, 312, 48, (0, 6), 20000),
    (361, 275, 32, (1, 8), 40000),     # not multiples of 8: cropped x8 maps
    (96, 96, 64, (0, 3), 4000),
    (656, 656, 16, (5, 30), 200000),
    (184, 248, 48, (0, 5), 20000),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=6.0)
    ap.add_argument('--workers', type=int, default=12)
    ap.add_argument('--seed', type=int, default=90000)
    ap.add_argument('--multiscale', action='store_true')
    args = ap.parse_args()
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    from tests import golden_io as gio

    pool = mp.get_context('spawn').Pool(args.workers)
    if args.multiscale:
        soak_multiscale(args, pool)
        pool.close()
        return
    t_end = time.time() + 60.0 * args.minutes
    seed = args.seed
    tot = dict(images=0, peaks=0, conns=0, persons=0, regrown=0)
    rng = np.random.RandomState(args.seed)
    rounds = 0
    while time.time() < t_end:
        for (H, W, B, people, max_pairs) in GEOMETRIES:
            if time.time() >= t_end:
                break
            mode = 'D' if (rounds + H) % 2 == 0 else 'P'
            t1 = float(rng.choice([0.1, 0.1, 0.05, 0.2]))
            t2 = float(rng.choice([0.05, 0.05, 0.02, 0.1]))
            noise = float(rng.choice([0.01, 0.01, 0.03, 0.002]))
            heat, paf = synth.batch(B, H, W, people, seed0=seed, paf_order='xy' if mode == 'D' else 'yx', noise=noise)
            seed += B
            eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=mode, max_peaks_per_part=128, max_persons=128,
                               max_pairs_per_image=max_pairs)
            t0 = time.time()
            try:
                res = eng.decode_batch(heat, paf, out_hw=(H, W), thre1=t1, thre2=t2)
            except _ffi.DecodeOverflow:
                # low threshold + strong noise on a large map: more than 128 peaks of a part.  The library
                # says so; the caller's answer is a context with larger lists.
                eng.close()
                eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=mode, max_peaks_per_part=512, max_persons=256,
                                   max_pairs_per_image=4 * max_pairs)
                res = eng.decode_batch(heat, paf, out_hw=(H, W), thre1=t1, thre2=t2)
                tot['regrown'] += 1
            t_gpu = time.time() - t0
            t0 = time.time()
            want = pool.map(oracle_image, [(heat[b], paf[b], mode, H, W, t1, t2) for b in range(B)])
            t_cpu = time.time() - t0
            n = dict(peaks=0, conns=0, persons=0)
            for b in range(B):
                peaks, conns, special, subset = want[b]
                try:
                    gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
                    g_conns, g_special = res.connections(b)
                    assert g_special == special
                    gio.assert_conns_equal(g_conns, conns, RTOL)
                    gio.assert_persons_equal(res.subset(b), subset, RTOL)
                except AssertionError as e:
                    print('MISMATCH: geometry %dx%d mode %s seed %d image %d thresholds %g %g noise %g: %s'
                          % (H, W, mode, seed - B, b, t1, t2, noise, e), flush=True)
                    sys.exit(1)
                n['peaks'] += sum(len(p) for p in peaks)
                n['conns'] += sum(len(c) for c in conns)
                n['persons'] += len(subset)
            eng.close()
            tot['images'] += B
            for k in n:
                tot[k] += n[k]
            if max(len(p) for b in range(B) for p in want[b][0]) > 128:
                assert eng.ctx.cfg.max_peaks_per_part == 512, 'more than 128 peaks of a part and no overflow reported'
            print('%4dx%-4d mode %s thr %.2f/%.2f noise %.3f: %3d images equal (%5d peaks, %5d connections, %4d persons); '
                  'decode %.3f s, oracle %.1f s' % (H, W, mode, t1, t2, noise, B, n['peaks'], n['conns'], n['persons'],
                                                   t_gpu, t_cpu), flush=True)
        rounds += 1
    pool.close()
    print('TOTAL: %(images)d images, %(peaks)d peaks, %(conns)d connections, %(persons)d persons compared with the oracle: '
          'all equal (%(regrown)d batches decoded again with larger lists after PD_E_OVERFLOW)' % tot)


if __name__ == '__main__':
    main()
