"""Differential fuzzing of the whole decode against the CPU oracle on inputs no scene generator makes:
smooth random heatmaps (a dozen maxima per part, of both signs) and smooth random vector fields as PAFs
(many candidate pairs pass the tests, peaks are shared between connections, persons merge, three owners
meet), at random small geometries with and without a crop, in both modes.  Every stage is compared (peaks
with exact scores, connection lists, person rows); where the reference raises IndexError (predict.py:199)
the library must raise it for the same batch.

With --multiscale: two to four random test scales between 0.4 and 2.2 with non-zero pads, independent random
network outputs per scale, through pd_decode_multiscale (no stored maps) against the oracle's resize + average +
decode (predict.py:49-254).

Usage (on a GPU box): python tools/fuzz_decode.py [--cases 200] [--seed 0] [--batch 4] [--multiscale | --large | --images | --shapes]
Prints one line per mismatch and a summary; exit status 1 if anything differed."""
import argparse
import os
import sys
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import decode_ref as ref                      # noqa: E402
from poseestimation_amd import predict_decode as pdec     # noqa: E402
from tests import golden_io as gio                        # noqa: E402


def smooth(rng, shape, passes):
    a = rng.standard_normal(shape).astype(np.float32)
    for _ in range(passes):
        a = (a + np.roll(a, 1, 0) + np.roll(a, -1, 0) + np.roll(a, 1, 1) + np.roll(a, -1, 1)) / np.float32(5)
    return a / np.float32(np.abs(a).max() + 1e-6)


def make_case(rng, large=False):
    if large:   # maps of more than 4096 pixels: the two-kernel peak stage, the layout copies, the pair pre-test
        h, w = int(rng.integers(60, 92)), int(rng.integers(70, 92))
        H, W = 8 * h - int(rng.integers(0, 8)), 8 * w - int(rng.integers(0, 8))
        heat = smooth(rng, (h, w, 19), int(rng.integers(12, 40))) * np.float32(rng.choice([0.6, 1.0]))
        paf = smooth(rng, (h, w, 38), int(rng.integers(4, 12))) * np.float32(0.5)
        return heat, paf, (H, W)
    h, w = int(rng.integers(4, 18)), int(rng.integers(4, 18))
    crop = rng.random() < 0.5
    H = 8 * h - (int(rng.integers(0, 8)) if crop else 0)
    W = 8 * w - (int(rng.integers(0, 8)) if crop else 0)
    amp = np.float32(rng.choice([0.3, 0.6, 1.0, 3.0]))
    heat = smooth(rng, (h, w, 19), int(rng.integers(0, 3))) * amp
    if rng.random() < 0.3:
        heat = np.abs(heat)
    paf = smooth(rng, (h, w, 38), int(rng.integers(1, 4))) * np.float32(rng.choice([0.2, 0.5, 1.0]))
    if rng.random() < 0.5:      # a coherent field: most pairs along it pass the count test
        ang = rng.uniform(0, 2 * np.pi)
        paf[..., 0::2] += np.float32(0.4 * np.cos(ang))
        paf[..., 1::2] += np.float32(0.4 * np.sin(ang))
    return heat, paf, (H, W)


def oracle(heat, paf, H, W, mode, t1, t2):
    hf = ref.resize_cubic(heat, fx=8, fy=8)[:H, :W].astype(np.float64)
    pf = ref.resize_cubic(paf, fx=8, fy=8)[:H, :W].astype(np.float64)
    peaks = ref.get_peaks(hf, t1, mode, use_scipy=True)
    conns, special = ref.get_connections(peaks, pf, t2, mode)
    try:
        subset = ref.multi_person_parse(peaks, conns, special, mode)
    except IndexError:
        subset = IndexError
    return peaks, conns, special, subset


def multiscale_cases(args):
    bad = retried = 0
    totals = [0, 0, 0]
    for case in range(args.cases):
        rng = np.random.default_rng(args.seed * 100003 + case + 7)
        H, W = int(rng.integers(40, 168)), int(rng.integers(40, 168))
        n = int(rng.integers(2, 5))
        scales = sorted(float(x) for x in rng.uniform(0.4, 2.2, n).round(2))
        mode = 'DP'[case % 2]
        B = args.batch
        amp = np.float32(rng.choice([0.3, 0.6, 1.0]))
        heats, pafs, pads = [], [], []
        for sc in scales:
            sh, sw = int(round(H * sc)), int(round(W * sc))
            hs, ws = -(-sh // 8), -(-sw // 8)
            heats.append(np.stack([smooth(rng, (hs, ws, 19), int(rng.integers(0, 3))) * amp for _ in range(B)]))
            pafs.append(np.stack([smooth(rng, (hs, ws, 38), int(rng.integers(1, 4))) * np.float32(0.6) for _ in range(B)]))
            pads.append((hs * 8 - sh, ws * 8 - sw))
        tag = 'multiscale case %d (seed %d) mode %s %dx%d scales %s' % (case, args.seed, mode, H, W, scales)
        eng = pdec.DecodeEngine(B, H, W, mode=mode, max_peaks_per_part=256, max_persons=204, max_pairs_per_image=400000)
        try:
            try:
                res = eng.decode_multiscale(heats, pafs, pads, (H, W))
            except pdec._ffi.DecodeOverflow:      # as above: again with larger tables
                eng.close()
                retried += 1
                eng = pdec.DecodeEngine(B, H, W, mode=mode, max_peaks_per_part=1024, max_persons=960, max_pairs_per_image=4000000)
                res = eng.decode_multiscale(heats, pafs, pads, (H, W))
        except IndexError:
            eng.close()
            continue
        except Exception as exc:
            bad += 1
            print(tag, ': decode raised', repr(exc))
            eng.close()
            continue
        for b in range(B):
            want_h, want_p = ref.upsample_average([h[b] for h in heats], [p[b] for p in pafs], pads, (H, W))
            try:
                peaks = ref.get_peaks(want_h, 0.1, mode, use_scipy=True)
                conns, special = ref.get_connections(peaks, want_p, 0.05, mode)
                subset = ref.multi_person_parse(peaks, conns, special, mode)
            except IndexError:
                continue
            try:
                gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
                g_conns, g_special = res.connections(b)
                assert g_special == special, ('special', g_special, special)
                gio.assert_conns_equal(g_conns, conns, 1e-9)
                gio.assert_persons_equal(res.subset(b), subset, 1e-9)
                totals[0] += sum(len(p) for p in peaks)
                totals[1] += sum(len(c) for c in conns)
                totals[2] += len(subset)
            except AssertionError as exc:
                bad += 1
                print(tag, 'image', b, ':', str(exc)[:400].replace('\n', ' | '))
        eng.close()
    print('%d multi-scale cases x %d images: %d mismatches; %d decoded again with larger tables after a capacity overflow; '
          'compared %d peaks, %d connections, %d persons' % (args.cases, args.batch, bad, retried, totals[0], totals[1], totals[2]))
    return 1 if bad else 0


def image_cases(args):
    """pd_image_resize_pad (predict.py:56-57) against the restatement: random sizes, scales 0.2..3, uint8 (noise,
    black / white blocks whose cubic overshoot saturates, constant) and float32."""
    bad = n = 0
    for case in range(args.cases):
        rng = np.random.default_rng(args.seed * 100003 + 1000 + case)
        ih, iw = int(rng.integers(9, 300)), int(rng.integers(9, 300))
        scale = float(rng.uniform(0.2, 3.0))
        if case % 2 == 0:
            kind = (case // 2) % 3
            if kind == 0:
                img = rng.integers(0, 256, (ih, iw, 3)).astype(np.uint8)
            elif kind == 1:
                img = (rng.integers(0, 2, (ih // 3 + 1, iw // 3 + 1, 3)) * 255).astype(np.uint8).repeat(3, 0).repeat(3, 1)[:ih, :iw]
            else:
                img = np.full((ih, iw, 3), int(rng.integers(0, 256)), np.uint8)
        else:
            img = (rng.random((ih, iw, 3)) * 255).astype(np.float32)
        img = np.ascontiguousarray(img)
        want, wpad = ref.image_resize_pad(img, scale, 8, 128)
        got, gpad = pdec.resize_pad(img, scale)
        n += 1
        if got.dtype != want.dtype or got.shape != want.shape or tuple(gpad) != tuple(wpad) or not np.array_equal(got, want):
            bad += 1
            print('image case %d (seed %d): %dx%d %s scale %.4f differs' % (case, args.seed, ih, iw, img.dtype, scale))
    print('%d image resize cases: %d mismatches' % (n, bad))
    return 1 if bad else 0


def shape_cases(args):
    """Degenerate geometries: maps of one to three network pixels a side, strips one to five pixels wide and up to 200
    long, with random crops; uniform noise (so every part has peaks wherever the map allows one)."""
    bad = n = 0
    shapes = [(1, 1), (1, 2), (2, 1), (2, 2), (1, 3), (3, 3), (1, 40), (40, 1), (2, 64), (64, 2), (3, 127), (127, 3), (1, 128), (128, 1),
              (5, 200), (200, 5)]
    for case in range(args.cases):
        h, w = shapes[case % len(shapes)]
        rng = np.random.default_rng(args.seed * 100003 + 5000 + case)
        H, W = 8 * h - int(rng.integers(0, 8)), 8 * w - int(rng.integers(0, 8))
        heat = rng.random((2, h, w, 19)).astype(np.float32)
        paf = rng.random((2, h, w, 38)).astype(np.float32) - np.float32(0.3)
        mode = 'DP'[case % 2]
        tag = 'shape case %d (seed %d): maps %dx%d -> %dx%d mode %s' % (case, args.seed, h, w, H, W, mode)
        eng = pdec.DecodeEngine(max_batch=2, max_h=8 * h, max_w=8 * w, mode=mode, max_peaks_per_part=512, max_persons=960,
                                max_pairs_per_image=1000000)
        try:
            res = eng.decode_batch(heat, paf, out_hw=(H, W))
        except IndexError:
            eng.close()
            continue
        except Exception as exc:
            bad += 1
            print(tag, ': decode raised', repr(exc))
            eng.close()
            continue
        for b in range(2):
            peaks, conns, special, subset = oracle(heat[b], paf[b], H, W, mode, 0.1, 0.05)
            if subset is IndexError:
                continue
            try:
                gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
                g_conns, g_special = res.connections(b)
                assert g_special == special, ('special', g_special, special)
                gio.assert_conns_equal(g_conns, conns, 1e-9)
                gio.assert_persons_equal(res.subset(b), subset, 1e-9)
                n += 1
            except AssertionError as exc:
                bad += 1
                print(tag, 'image', b, ':', str(exc)[:400].replace('\n', ' | '))
        eng.close()
    print('%d shape cases: %d images equal, %d mismatches' % (args.cases, n, bad))
    return 1 if bad else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', type=int, default=200)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--multiscale', action='store_true')
    ap.add_argument('--images', action='store_true', help='the image resize + pad in front of the network instead of the decode')
    ap.add_argument('--shapes', action='store_true', help='maps of one to three pixels a side and long strips')
    ap.add_argument('--impl', type=int, default=0, help='peak-stage implementation (pd_set_impl): 0 candidate filter, 1 per pixel, 2 exact streaming')
    ap.add_argument('--large', action='store_true', help='maps of 60..91 x 70..91 network pixels')
    args = ap.parse_args()
    warnings.simplefilter('ignore')
    if args.multiscale:
        return multiscale_cases(args)
    if args.images:
        return image_cases(args)
    if args.shapes:
        return shape_cases(args)
    bad = 0
    index_errors = retried = 0
    totals = [0, 0, 0]
    for case in range(args.cases):
        rng = np.random.default_rng(args.seed * 100003 + case)
        heat0, paf0, (H, W) = make_case(rng, args.large)
        mode = 'DP'[case % 2]
        t1, t2 = float(rng.choice([0.1, 0.05, 0.3])), float(rng.choice([0.05, 0.02, 0.2]))
        # the images of a batch share the geometry: the first is the case, the others are its flips and negation
        imgs = [(heat0, paf0), (heat0[::-1].copy(), paf0[::-1].copy()), (heat0[:, ::-1].copy(), paf0[:, ::-1].copy()),
                (-heat0, -paf0)][:args.batch]
        want = [oracle(hm, pm, H, W, mode, t1, t2) for hm, pm in imgs]
        eng = pdec.DecodeEngine(max_batch=len(imgs), max_h=8 * heat0.shape[0], max_w=8 * heat0.shape[1], mode=mode,
                                max_peaks_per_part=512 if args.large else 256, max_persons=960 if args.large else 204, max_pairs_per_image=2000000 if args.large else 400000)
        eng.ctx.set_impl(args.impl)
        tag = 'case %d (seed %d) mode %s %dx%d -> %dx%d thr %.2f/%.2f' % (case, args.seed, mode, heat0.shape[0], heat0.shape[1],
                                                                         H, W, t1, t2)
        expect_index = [b for b, wv in enumerate(want) if wv[3] is IndexError]
        try:
            res = eng.decode_batch(np.stack([i[0] for i in imgs]), np.stack([i[1] for i in imgs]), out_hw=(H, W),
                                   thre1=t1, thre2=t2)
        except IndexError as exc:
            index_errors += 1
            if not expect_index:
                bad += 1
                print(tag, ': the library raised IndexError, the oracle did not:', exc)
            eng.close()
            continue
        except pdec._ffi.DecodeOverflow:
            # more peaks, pairs or open persons than this context's tables hold (reported, never truncated): the same
            # batch again through a context with larger tables, which takes the separate assembly and packing kernels
            eng.close()
            retried += 1
            eng = pdec.DecodeEngine(max_batch=len(imgs), max_h=8 * heat0.shape[0], max_w=8 * heat0.shape[1], mode=mode,
                                    max_peaks_per_part=1024, max_persons=960, max_pairs_per_image=4000000)
            try:
                res = eng.decode_batch(np.stack([i[0] for i in imgs]), np.stack([i[1] for i in imgs]), out_hw=(H, W),
                                       thre1=t1, thre2=t2)
            except IndexError:
                index_errors += 1
                if not expect_index:
                    bad += 1
                    print(tag, ': the library raised IndexError, the oracle did not')
                eng.close()
                continue
        except Exception as exc:
            bad += 1
            print(tag, ': decode raised', repr(exc))
            eng.close()
            continue
        if expect_index:
            bad += 1
            print(tag, ': the oracle raised IndexError for images', expect_index, 'the library did not')
            eng.close()
            continue
        for b, (peaks, conns, special, subset) in enumerate(want):
            try:
                gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
                g_conns, g_special = res.connections(b)
                assert g_special == special, ('special', g_special, special)
                gio.assert_conns_equal(g_conns, conns, 1e-9)
                gio.assert_persons_equal(res.subset(b), subset, 1e-9)
                totals[0] += sum(len(p) for p in peaks)
                totals[1] += sum(len(c) for c in conns)
                totals[2] += len(subset)
            except AssertionError as exc:
                bad += 1
                print(tag, 'image', b, ':', str(exc)[:400].replace('\n', ' | '))
        eng.close()
    print('%d cases x %d images: %d mismatches; %d batches raised IndexError on both sides, %d decoded again with larger tables '
          'after a capacity overflow; compared %d peaks, %d connections, %d persons'
          % (args.cases, args.batch, bad, index_errors, retried, totals[0], totals[1], totals[2]))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
