"""The entry points either side of the decode, through the C ABI: COCO keypoint rows
(evaluate.py:44-74), the image side of the scale loop (predict.py:56-57), detect() end to
end with a stand-in network, rendering (predict.py:256-307), the limb-18 debug lines
(utils/postprocess.py:224-225) and the framework-free gather (one rank).

cv2 and pycocotools are not importable here and the reference holds no fixture for these
steps: expected values are derived by hand from the source text or come from the CPU
restatement in oracle/decode_ref.py (parity unpinned for the cubic and the rasteriser)."""
import numpy as np
import pytest

from oracle import decode_ref as ref

pytestmark = pytest.mark.gpu

COCO_TO_OURS = [0, 15, 14, 17, 16, 5, 2, 6, 3, 7, 4, 11, 8, 12, 9, 13, 10]   # cfgs/config_vgg19.py:48


def _peak(pid):
    # peak id i sits at y = 10 + i, x = 20 + 2 i
    return (np.int64(10 + pid), np.int64(20 + 2 * pid), np.float64(0.75), pid)


def _two_person_case():
    """Persons X = parts {1, 2, 3, 16} with peak ids {1, 3, 4, 7} and Y = parts {0, 1, 14, 15}
    with ids {0, 2, 5, 6}; every other part is missing."""
    peaks = [[] for _ in range(18)]
    peaks[0] = [_peak(0)]
    peaks[1] = [_peak(1), _peak(2)]
    peaks[2] = [_peak(3)]
    peaks[3] = [_peak(4)]
    peaks[14] = [_peak(5)]
    peaks[15] = [_peak(6)]
    peaks[16] = [_peak(7)]
    rows = {6: [[1, 3]], 7: [[3, 4]], 9: [[3, 7]], 14: [[2, 0]], 15: [[0, 5]], 16: [[0, 6]]}
    special = [k for k in range(19) if k not in rows]
    conns = [np.array([[a, b, 0.5, 0.75, 0.75] for a, b in rows[k]]) if k in rows else [] for k in range(19)]
    return peaks, conns, special


@pytest.mark.parametrize('mode', ['D', 'P'])
def test_coco_rows_known_answer(mode):
    """evaluate.py:44-74 by hand: keypoint k of a person is (candidate[id][0], candidate[id][1], 1)
    for id = person[coco_to_ours[k]], (0, 0, 0) where the part is missing; candidate's first two
    columns are (x, y) in predict.detect and (y, x) for utils/postprocess.py tuples."""
    from poseestimation_amd import _ffi, postprocess as pp
    peaks, conns, special = _two_person_case()
    with _ffi.Context(max_batch=1, max_h=64, max_w=64, max_peaks_per_part=16, max_persons=16,
                      max_pairs_per_image=64) as ctx:
        ctx.set_mode(_ffi.MODE_D if mode == 'D' else _ffi.MODE_P)
        ctx.load_fullres(None, np.zeros((1, 64, 64, 38)), 1, 64, 64)
        pp._upload_peaks(ctx, peaks)
        pp._upload_connections(ctx, peaks, conns, special)
        ctx.parse()
        kp, score, image = ctx.coco_rows()
    want = np.zeros((2, 17, 3))
    for row, parts in enumerate([{2: 3, 3: 4, 16: 7}, {0: 0, 14: 5, 15: 6}]):   # part -> peak id (part 1 is not a COCO keypoint)
        for part, pid in parts.items():
            y, x = 10 + pid, 20 + 2 * pid
            want[row, COCO_TO_OURS.index(part)] = (x, y, 1) if mode == 'D' else (y, x, 1)
    np.testing.assert_array_equal(kp, want)
    np.testing.assert_array_equal(score, [4.5, 4.5])      # 0.75 + 0.75 + 0.5, then twice + 0.75 + 0.5
    np.testing.assert_array_equal(image, [0, 0])


def test_coco_rows_of_a_batch_match_the_restatement():
    from poseestimation_amd import coco, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    heat, paf = synth.batch(6, H, W, (0, 4), seed0=300, paf_order='xy')
    eng = DecodeEngine(max_batch=6, max_h=H, max_w=W, mode='D')
    res = eng.decode_batch(heat, paf)
    ids = [1000 + 7 * b for b in range(6)]
    got = coco.device_records(eng.ctx, ids)
    want = []
    for b in range(6):
        want.extend(ref.coco_records(res.candidate(b), res.subset(b), ids[b], COCO_TO_OURS))
    assert len(got) == len(want) > 0
    for g, w in zip(got, want):
        assert g['image_id'] == w['image_id'] and g['category_id'] == 1
        assert g['keypoints'] == w['keypoints'] and g['score'] == w['score']
    assert got == coco.batch_records(res, ids)
    eng.close()


@pytest.mark.parametrize('dtype', [np.uint8, np.float32])
@pytest.mark.parametrize('h,w,scale', [(97, 131, 0.5 * 368 / 97), (97, 131, 368 / 97.0), (97, 131, 2.0 * 368 / 97),
                                       (64, 48, 1.0), (50, 70, 0.37), (33, 35, 4.1)])
def test_image_resize_pad_matches_restatement(dtype, h, w, scale):
    """predict.py:56-57: cubic resize by fx = fy = scale, padding below / right to a multiple of 8
    with 128; bit for bit against the float32 restatement of cv2.resize, uint8 rounded half to
    even and saturated."""
    from poseestimation_amd import predict_decode as pdec
    rng = np.random.RandomState(3)
    img = rng.randint(0, 256, size=(h, w, 3)).astype(dtype)
    if dtype == np.float32:
        img += rng.rand(h, w, 3).astype(np.float32)
    got, pad = pdec.resize_pad(img, scale)
    want, want_pad = ref.image_resize_pad(img, scale, 8, 128)
    assert pad == want_pad and got.dtype == want.dtype and got.shape == want.shape
    assert got.shape[0] % 8 == 0 and got.shape[1] % 8 == 0
    np.testing.assert_array_equal(got, want)
    rh, rw = got.shape[0] - pad[2], got.shape[1] - pad[3]
    assert (rh, rw) == (int(round(h * scale)), int(round(w * scale)))
    assert np.all(got[rh:] == 128) and np.all(got[:, rw:] == 128)


def _stub_network(imgs):
    """Stands in for the OfflinePredictor of predict.py:34-43: deterministic smooth maps of the
    stride-8 size, a function of the padded image."""
    from scipy.ndimage import gaussian_filter
    _, ph, pw, _ = imgs.shape
    hs, ws = ph // 8, pw // 8
    rng = np.random.RandomState(hs * 1000 + ws)
    base = imgs[0, 4::8, 4::8, :].astype(np.float32).mean(axis=2)[:hs, :ws] / 255.0
    def blobs(sigma):
        g = gaussian_filter(rng.randn(hs, ws), sigma)
        return g / g.std()
    heat = np.stack([0.3 * blobs(hs / 10.0) + 0.15 + 0.1 * base for _ in range(19)], axis=2)
    paf = np.stack([0.5 * blobs(hs / 12.0) for _ in range(38)], axis=2)
    return heat[None].astype(np.float32), paf[None].astype(np.float32)


def test_detect_end_to_end_with_a_stand_in_network():
    """predict.detect's whole loop (predict.py:45-254) with the image resize, the map resizes,
    the average and the decode on the device, against the same loop on the CPU restatement."""
    from poseestimation_amd import predict_decode as pdec
    from tests import golden_io as gio
    rng = np.random.RandomState(12)
    img = rng.randint(0, 256, size=(120, 150, 3)).astype(np.uint8)
    cand, peaks, subset = pdec.detect(img, _stub_network)
    h, w = img.shape[:2]
    heats, pafs, pads = [], [], []
    for x in (0.5, 1, 1.5, 2):
        padded, pad = ref.image_resize_pad(img, x * 368 / h, 8, 128)
        hm, pf = _stub_network(padded[None])
        heats.append(hm[0])
        pafs.append(pf[0])
        pads.append((pad[2], pad[3]))
    heat_avg, paf_avg = ref.upsample_average(heats, pafs, pads, (h, w))
    want_cand, want_peaks, want_subset = ref.decode(heat_avg, paf_avg, 0.1, 0.05, 'D')
    assert 10 < sum(len(p) for p in want_peaks) < 600
    gio.assert_peaks_equal(peaks, want_peaks, exact_score=True)
    np.testing.assert_array_equal(cand, want_cand)
    gio.assert_persons_equal(subset, want_subset, 1e-5)


def test_render_matches_the_restated_rasteriser():
    """predict.py:256-307 against the oracle's restatement of cv2.circle / ellipse2Poly /
    fillConvexPoly / addWeighted: both images bit for bit, also with limbs that leave the image
    on every side and with limbs of no length."""
    from poseestimation_amd import predict_decode as pdec, synth
    H, W = 184, 248
    heat, paf = synth.scene(H, W, 3, 41, paf_order='xy')
    eng = pdec.DecodeEngine(max_batch=1, max_h=H, max_w=W, mode='D')
    cand, peaks, subset = eng.decode_batch(heat[None], paf[None], out_hw=(H, W)).image(0)
    eng.close()
    assert len(subset) >= 2
    img = np.random.RandomState(1).randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    part, final = pdec.visualize(img, cand, peaks, subset, write=False)
    want_part, want_final = ref.visualize(img, cand, peaks, subset)
    np.testing.assert_array_equal(part, want_part)
    np.testing.assert_array_equal(final, want_final)
    assert np.any(final != img) and np.any(part != img)
    # made-up skeletons: every angle, limbs across the borders, coincident ends
    rng = np.random.RandomState(2)
    n_cand = 18 * 6
    cand2 = np.zeros((n_cand, 4))
    cand2[:, 0] = rng.randint(-30, W + 30, n_cand)
    cand2[:, 1] = rng.randint(-30, H + 30, n_cand)
    cand2[:6, :2] = cand2[6:12, :2]                      # parts 0 and 1 of every person coincide
    cand2[:, 3] = np.arange(n_cand)
    peaks2 = [[(int(cand2[p * 6 + j, 0]), int(cand2[p * 6 + j, 1]), 0.5, p * 6 + j) for j in range(6)] for p in range(18)]
    subset2 = -np.ones((6, 20))
    for j in range(6):
        subset2[j, :18] = np.arange(18) * 6 + j
    subset2[2, 4] = -1
    part2, final2 = pdec.visualize(img, cand2, peaks2, subset2, write=False)
    want_part2, want_final2 = ref.visualize(img, cand2, peaks2, subset2)
    np.testing.assert_array_equal(part2, want_part2)
    np.testing.assert_array_equal(final2, want_final2)
    # nothing to draw: the image comes back unchanged
    p0, f0 = pdec.visualize(img, np.zeros((0, 4)), [[] for _ in range(18)], np.zeros((0, 20)), write=False)
    np.testing.assert_array_equal(f0, img)
    np.testing.assert_array_equal(p0, img)


def test_limb18_debug_lines_behind_a_flag(capsys, monkeypatch):
    """utils/postprocess.py:224-225 prints `connection, len(found)` for every connection of limb
    18 (parts 15 -> 17).  Here: person A holds part 15 (id 6) when limb 18 arrives, so the first
    connection finds one person; the second connection's ends belong to nobody (and limb 18
    creates no person): zero."""
    from poseestimation_amd import postprocess as pp
    peaks, conns, special = _two_person_case()
    peaks[15] = [_peak(6), _peak(8)]
    peaks[17] = [_peak(9), _peak(10)]
    conns[18] = np.array([[6, 9, 0.5, 0.75, 0.75], [8, 10, 0.25, 0.75, 0.75]])
    special = [k for k in special if k != 18]
    quiet = pp.multi_person_parse(peaks, conns, special)
    assert capsys.readouterr().out == ''
    monkeypatch.setattr(pp, 'debug_print', True)
    loud = pp.multi_person_parse(peaks, conns, special)
    out = capsys.readouterr().out
    assert out == '%s 1\n%s 0\n' % (conns[18][0], conns[18][1])
    np.testing.assert_array_equal(quiet, loud)
    y = loud[1]
    assert y[17] == 9 and y[19] == 5        # the first connection added part 17 to person Y


def test_gather_through_the_c_abi_one_rank():
    """pd_comm_* / pd_gather with RCCL loaded by the library itself: one rank, two contexts; the
    gathered buffers are the contexts' skeleton buffers, in context order."""
    import torch
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    engs, bufs, want = [], [], []
    for i in range(2):
        heat, paf = synth.batch(3 + i, H, W, (1, 4), seed0=500 + 10 * i, paf_order='xy')
        eng = DecodeEngine(max_batch=4, max_h=H, max_w=W, mode='D')
        buf = torch.zeros((1 << 16,), dtype=torch.uint8, device='cuda')
        eng.ctx.set_skeleton_buffer(buf.data_ptr(), buf.numel())
        want.append(eng.decode_batch(heat, paf).packed)
        engs.append(eng)
        bufs.append(buf)
    with _ffi.Comm(_ffi.Comm.unique_id(), 1, 0, 0) as comm:
        blobs = comm.gather([e.ctx for e in engs], root=0, capacity=1 << 18)
        assert len(blobs) == 2
        for blob, w in zip(blobs, want):
            got = _ffi.unpack_results(blob.copy())
            for key in ('part_count', 'person_count', 'status', 'peaks', 'persons'):
                assert np.array_equal(got[key], w[key]), key
            assert len(got['conns']) == 0
        with pytest.raises(_ffi.DecodeOverflow):
            comm.gather([e.ctx for e in engs], root=0, capacity=64)
    for e in engs:
        e.close()


_GATHER_WORLD_SCRIPT = r'''
import sys, threading
import numpy as np
sys.path.insert(0, %r)
import torch
from poseestimation_amd import _ffi, synth
from poseestimation_amd.predict_decode import DecodeEngine

WORLD, NCTX, ROOT, H, W, B = 3, 2, 1, 184, 184, 3
uid = _ffi.Comm.unique_id()
comms = [_ffi.Comm(uid, WORLD, r, 0) for r in range(WORLD)]
engs, bufs, want = {}, {}, {}
plain = DecodeEngine(B, H, W, mode='D')
for r in range(WORLD):
    for i in range(NCTX):
        heat, paf = synth.batch(B, H, W, (1, 4), seed0=1000 + 10 * r + i, paf_order='xy')
        want[r, i] = plain.decode_batch(heat, paf).packed
        e = DecodeEngine(B, H, W, mode='D')
        buf = torch.zeros(1 << 16, dtype=torch.uint8, device='cuda')
        e.ctx.set_skeleton_buffer(buf.data_ptr(), buf.numel())
        e.enqueue(heat, paf)
        engs[r, i], bufs[r, i] = e, buf


def gather_all(capacity=1 << 20, ctxs=None):
    out, err = [None] * WORLD, [None] * WORLD

    def run(r):
        try:
            out[r] = comms[r].gather(ctxs[r] if ctxs else [engs[r, i].ctx for i in range(NCTX)], root=ROOT, capacity=capacity)
        except Exception as exc:
            err[r] = exc
    ts = [threading.Thread(target=run, args=(r,)) for r in range(WORLD)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(60)
        assert not t.is_alive(), 'a rank is stuck in pd_gather'
    return out, err


# 1. three ranks, two contexts each, a root that is not rank 0: rank-major, then context order
out, err = gather_all()
assert err == [None] * WORLD, err
assert out[0] is None and out[2] is None and len(out[ROOT]) == WORLD * NCTX
for r in range(WORLD):
    for i in range(NCTX):
        got = _ffi.unpack_results(out[ROOT][r * NCTX + i])
        for key in ('part_count', 'person_count', 'status', 'peaks', 'persons'):
            assert np.array_equal(got[key], want[r, i][key]), (r, i, key)
        assert len(got['conns']) == 0
print('CASE1 ok')
# 2. a host array on the root that is too small: reported there, after the exchange; nobody hangs
out, err = gather_all(capacity=256)
assert isinstance(err[ROOT], _ffi.DecodeOverflow) and err[0] is None and err[2] is None, err
print('CASE2 ok')
# 3. a skeleton buffer that no decode has written (rank 2): every rank gives up together
fresh = DecodeEngine(B, H, W, mode='D')
fbuf = torch.zeros(1 << 16, dtype=torch.uint8, device='cuda')
fresh.ctx.set_skeleton_buffer(fbuf.data_ptr(), fbuf.numel())
ctxs = {r: [engs[r, i].ctx for i in range(NCTX)] for r in range(WORLD)}
ctxs[2] = [engs[2, 0].ctx, fresh.ctx]
out, err = gather_all(ctxs=ctxs)
assert all(isinstance(e, RuntimeError) for e in err), err
assert 'does not hold the results' in str(err[2]) and 'holds no decode results' in str(err[0]), err
print('CASE3 ok')
# 4. the exchange still works afterwards
out, err = gather_all()
assert err == [None] * WORLD and len(out[ROOT]) == WORLD * NCTX
print('CASE4 ok')
'''


def test_gather_protocol_with_three_ranks_over_a_fake_transport():
    """pd_gather with a world of three ranks inside one process (host threads, one GPU): RCCL is
    replaced by tests/fake_rccl (the entry points pd_gather uses, on device-to-device copies), so
    that the root's receives from other ranks, the rank-major order, a root other than rank 0
    and the error paths that must not leave a rank waiting all run.  A child process, because the
    library picks its RCCL once."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    fake = os.path.join(root, 'tests', 'fake_rccl', 'libfakerccl.so')
    if not os.path.exists(fake):
        pytest.skip('tests/fake_rccl/libfakerccl.so is not built (python -c "import __graft_entry__ as g; g.build()")')
    env = dict(os.environ, PD_RCCL_LIBRARY=fake)
    p = subprocess.run([sys.executable, '-c', _GATHER_WORLD_SCRIPT % root], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=200)
    assert p.returncode == 0, p.stdout[-3000:]
    for case in ('CASE1 ok', 'CASE2 ok', 'CASE3 ok', 'CASE4 ok'):
        assert case in p.stdout, p.stdout[-3000:]
