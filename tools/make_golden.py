#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's utils/postprocess.py.

Runs only in the build container (needs /root/reference).  The reference's files
are imported from where they lie and never copied; the only thing missing for the
import is the third-party ``easydict`` package, used by cfgs/config_*.py purely as
an attribute dict, so an in-memory attribute-dict module stands in for it.

Fixtures hold data only: inputs (quantised network outputs as int16 multiples of
2**-12, or small full-resolution float arrays) and the reference's outputs.
"""
import contextlib
import hashlib
import io
import os
import sys
import types
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True


def import_reference():
    class AttrDict(dict):
        def __getattr__(self, k):
            try:
                return self[k]
            except KeyError:
                raise AttributeError(k)

        def __setattr__(self, k, v):
            self[k] = v
    mod = types.ModuleType('easydict')
    mod.EasyDict = AttrDict
    sys.modules['easydict'] = mod
    sys.path.insert(0, '/root/reference')
    warnings.simplefilter('ignore')
    import utils.postprocess as pp
    return pp


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack_peaks(peaks):
    rows, part = [], []
    for p, plist in enumerate(peaks):
        for t in plist:
            rows.append([float(t[0]), float(t[1]), float(t[2]), float(t[3])])
            part.append(p)
    return np.array(rows, np.float64).reshape(-1, 4), np.array(part, np.int32)


def pack_conns(conns):
    rows, limb, is_list = [], [], []
    for k, c in enumerate(conns):
        is_list.append(isinstance(c, list))
        for r in c:
            rows.append(list(map(float, r)))
            limb.append(k)
    return (np.array(rows, np.float64).reshape(-1, 5), np.array(limb, np.int32),
            np.array(is_list, bool))


def run_reference(pp, heat_full, paf_full, thre1=0.1, thre2=0.05):
    peaks = pp.get_peaks(heat_full, thre1)
    conns, special = pp.get_connections(peaks, paf_full, thre2)
    with contextlib.redirect_stdout(io.StringIO()):       # stray print, postprocess.py:224-225
        persons = pp.multi_person_parse(peaks, conns, special)
    return peaks, conns, special, persons


def margins_ok(ref, peaks, paf_full, thre2, eps=1e-9):
    """Reject scenes whose discrete outcome hangs on the last bits of a PAF score
    (the reference uses numpy pow for the limb length, which is not correctly
    rounded; a different-but-valid sqrt must not flip a decision)."""
    for k in range(19):
        ca, cb = peaks[ref.LIMB_FROM[k]], peaks[ref.LIMB_TO[k]]
        if not ca or not cb:
            continue
        ia = np.repeat(np.arange(len(ca)), len(cb)); ib = np.tile(np.arange(len(cb)), len(ca))
        ya = np.array([p[0] for p in ca])[ia]; xa = np.array([p[1] for p in ca])[ia]
        yb = np.array([p[0] for p in cb])[ib]; xb = np.array([p[1] for p in cb])[ib]
        d2 = (yb - ya) ** 2 + (xb - xa) ** 2
        keep = d2 > 0
        if not keep.any():
            continue
        ya, xa, yb, xb = ya[keep], xa[keep], yb[keep], xb[keep]
        n = np.sqrt(((yb - ya) ** 2 + (xb - xa) ** 2).astype(np.float64))
        ly = ref._linspace10(ya, yb).astype(np.int64); lx = ref._linspace10(xa, xb).astype(np.int64)
        mid = (paf_full[ly, lx, 2 * k] * ((yb - ya) / n)[:, None]
               + paf_full[ly, lx, 2 * k + 1] * ((xb - xa) / n)[:, None])
        if np.any(np.abs(mid - thre2) < eps):
            return False
        score, ok = ref.pair_scores(ya, xa, yb, xb, paf_full[:, :, 2 * k:2 * k + 2],
                                    paf_full.shape[0], thre2, 'P')
        if np.any(np.abs(score[np.isfinite(score)]) < eps):
            return False
        s = np.sort(score[ok])
        if len(s) > 1 and np.any(np.diff(s) < eps):
            return False
    return True


def main():
    from oracle import decode_ref as ref
    from poseestimation_amd import synth
    pp = import_reference()
    out_dir = os.path.join(ROOT, 'tests', 'golden')
    os.makedirs(out_dir, exist_ok=True)

    # ---- A. scene fixtures through the whole mode-P pipeline -------------------
    scenes = [('scene368_p5', 368, 368, 5, 0), ('scene368_p3', 368, 368, 3, 1),
              ('scene368_p8', 368, 368, 8, 2), ('scene_184x248_p2', 184, 248, 2, 3),
              ('scene368_p0', 368, 368, 0, 4), ('scene656_p30', 656, 656, 30, 5)]
    for name, H, W, P, seed in scenes:
        while True:
            heat, paf = synth.scene(H, W, P, seed, paf_order='yx', quantize=True)
            heat_full = ref.resize_cubic(heat, fx=8, fy=8).astype(np.float64)
            paf_full = ref.resize_cubic(paf, fx=8, fy=8).astype(np.float64)
            peaks, conns, special, persons = run_reference(pp, heat_full, paf_full)
            if margins_ok(ref, peaks, paf_full, 0.05):
                break
            seed += 1000
            print(name, 're-seeding to', seed)
        pk, pk_part = pack_peaks(peaks)
        cn, cn_limb, cn_list = pack_conns(conns)
        np.savez_compressed(
            os.path.join(out_dir, name + '.npz'),
            H=H, W=W, P=P, seed=seed,
            heat_q=np.round(heat * 4096).astype(np.int16), paf_q=np.round(paf * 4096).astype(np.int16),
            heat_full_sha=sha(heat_full), paf_full_sha=sha(paf_full),
            peaks=pk, peaks_part=pk_part, conns=cn, conns_limb=cn_limb, conns_is_list=cn_list,
            special=np.array(special, np.int32), persons=np.asarray(persons, np.float64))
        print(name, 'peaks', len(pk), 'conns', len(cn), 'persons', len(persons))

    # ---- B. direct full-resolution get_peaks cases -----------------------------
    rng = np.random.RandomState(7)
    cases = {}
    plateau = np.zeros((64, 80, 18))
    plateau[10:50, 20:70, :] = 0.75                      # flat top: every interior pixel is a peak
    plateau[:, :, 1] = rng.rand(64, 80)
    plateau[0:3, 0:4, 2] = 1.9                           # peak touching the corner (zero-filled neighbours)
    cases['plateau'] = plateau
    cases['tiny_9x11'] = rng.rand(9, 11, 18)             # shorter than the filter radius: multiple reflection
    cases['f32_48x40'] = rng.rand(48, 40, 19).astype(np.float32)
    cases['neg_40x40'] = rng.randn(40, 40, 18) * 0.3
    for name, hm in cases.items():
        thr = -0.05 if name.startswith('neg') else 0.1
        q = np.round(hm * 16384.0).astype(np.int16)      # inputs are multiples of 2**-14
        hm = (q.astype(np.float64) / 16384.0).astype(hm.dtype)
        peaks = pp.get_peaks(hm, thr)
        pk, pk_part = pack_peaks(peaks)
        extra = {}
        if name in ('tiny_9x11', 'f32_48x40'):
            extra['smoothed'] = np.stack([pp.gaussian_filter(hm[:, :, c], sigma=3) for c in range(18)], axis=2)
        np.savez_compressed(os.path.join(out_dir, 'peaks_' + name + '.npz'),
                            heatmap_q=q, dtype=str(hm.dtype), threshold=thr, peaks=pk,
                            peaks_part=pk_part, **extra)
        print('peaks_' + name, len(pk))

    # ---- C. get_connections on hand-made peak lists -----------------------------
    for idx in range(4):
        rng = np.random.RandomState(100 + idx)
        H, W = 40, 56
        paf = (rng.randn(H, W, 38) * 0.4)
        if idx == 3:
            paf = paf.astype(np.float32)
        peaks, pid = [], 0
        for part in range(18):
            n = int(rng.randint(0, 6)) if idx != 2 else int(rng.randint(0, 2))
            plist = []
            for _ in range(n):
                y, x = np.int64(rng.randint(0, H)), np.int64(rng.randint(0, W))
                plist.append((y, x, np.float64(rng.rand()), pid))
                pid += 1
            peaks.append(plist)
        if idx == 0:                                     # coincident peaks: length == 0 is skipped
            peaks[8] = [(peaks[1][0][0], peaks[1][0][1], np.float64(0.5), 900)] if peaks[1] else peaks[8]
        # strong field along a few limbs so that some candidates survive
        for k in range(19):
            a, b = ref.LIMB_FROM[k], ref.LIMB_TO[k]
            for pa in peaks[a][:2]:
                for pb in peaks[b][:2]:
                    dy, dx = float(pb[0] - pa[0]), float(pb[1] - pa[1])
                    n = math_hypot(dy, dx)
                    if n == 0:
                        continue
                    ys = np.linspace(pa[0], pb[0], 60).astype(int); xs = np.linspace(pa[1], pb[1], 60).astype(int)
                    paf[ys, xs, 2 * k] = dy / n
                    paf[ys, xs, 2 * k + 1] = dx / n
        paf_q = np.round(paf * 4096.0).astype(np.int16)  # inputs are multiples of 2**-12
        paf = (paf_q.astype(np.float64) / 4096.0).astype(paf.dtype)
        conns, special = pp.get_connections(peaks, paf, 0.05)
        pk, pk_part = pack_peaks(peaks)
        cn, cn_limb, cn_list = pack_conns(conns)
        np.savez_compressed(os.path.join(out_dir, 'conn_case%d.npz' % idx), paf_q=paf_q,
                            dtype=str(paf.dtype), peaks=pk,
                            peaks_part=pk_part, conns=cn, conns_limb=cn_limb, conns_is_list=cn_list,
                            special=np.array(special, np.int32), threshold=0.05)
        print('conn_case%d' % idx, 'conns', len(cn), 'special', special)

    # ---- D. multi_person_parse on random connection structures ------------------
    all_cases = []
    for idx in range(200):
        rng = np.random.RandomState(5000 + idx)
        peaks, pid = [], 0
        for part in range(18):
            n = int(rng.randint(0, 5))
            plist = [(np.int64(rng.randint(0, 100)), np.int64(rng.randint(0, 100)),
                      np.float64(rng.rand()), pid + i) for i in range(n)]
            pid += n
            peaks.append(plist)
        conns, special = [], []
        for k in range(19):
            ca, cb = peaks[ref.LIMB_FROM[k]], peaks[ref.LIMB_TO[k]]
            if len(ca) * len(cb) == 0:
                special.append(k); conns.append([]); continue
            m = int(rng.randint(0, min(len(ca), len(cb)) + 1))
            ia = rng.permutation(len(ca))[:m]; ib = rng.permutation(len(cb))[:m]
            sc = np.sort(rng.rand(m) * 1.5)[::-1]
            rows = [[ca[i][3], cb[j][3], s, ca[i][2], cb[j][2]] for i, j, s in zip(ia, ib, sc)]
            conns.append(np.array(rows, np.float64).reshape(-1, 5))
        with contextlib.redirect_stdout(io.StringIO()):
            persons = pp.multi_person_parse(peaks, conns, special)
        pk, pk_part = pack_peaks(peaks)
        cn, cn_limb, cn_list = pack_conns(conns)
        all_cases.append(dict(peaks=pk, peaks_part=pk_part, conns=cn, conns_limb=cn_limb,
                              special=np.array(special, np.int32), persons=np.asarray(persons, np.float64)))
    flat = {}
    for i, c in enumerate(all_cases):
        for key, v in c.items():
            flat['%03d_%s' % (i, key)] = v
    np.savez_compressed(os.path.join(out_dir, 'parse_cases.npz'), n=len(all_cases), **flat)
    print('parse_cases', len(all_cases), 'persons total', sum(len(c['persons']) for c in all_cases))


def math_hypot(a, b):
    import math
    return math.hypot(a, b)


if __name__ == '__main__':
    main()
