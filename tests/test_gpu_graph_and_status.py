"""Launch-graph invalidation, error reporting on the skeleton-only path, the device-resident
hand-off against the oracle, and the result entry points that copy into caller arrays."""
import numpy as np
import pytest

from oracle import decode_ref as ref
from tests import golden_io as gio

pytestmark = pytest.mark.gpu

RTOL = 1e-5
KEYS = ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons')


def _oracle(heat, paf, H, W, mode='D'):
    hf = ref.resize_cubic(heat, fx=8, fy=8)[:H, :W].astype(np.float64)
    pf = ref.resize_cubic(paf, fx=8, fy=8)[:H, :W].astype(np.float64)
    peaks = ref.get_peaks(hf, 0.1, mode, use_scipy=True)
    conns, special = ref.get_connections(peaks, pf, 0.05, mode)
    return peaks, conns, special, ref.multi_person_parse(peaks, conns, special, mode)


def test_replayed_graph_is_dropped_when_another_call_moves_buffers_or_tables():
    """K1 three times (recorded and replayed), then K2 with a larger batch and another
    crop height through the same context (it reallocates the staging buffers and rewrites
    the tap and filter tables), then K1 again: every result must equal an eager context's."""
    import torch
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    h1, p1 = synth.batch(2, 368, 368, (2, 5), seed0=500, paf_order='xy')
    h2, p2 = synth.batch(5, 368, 368, (2, 5), seed0=510, paf_order='xy')
    d1 = (torch.from_numpy(h1).cuda(), torch.from_numpy(p1).cuda())
    d2 = (torch.from_numpy(h2).cuda(), torch.from_numpy(p2).cuda())
    eager = DecodeEngine(max_batch=5, max_h=368, max_w=368, mode='D')
    eager.ctx.set_graph_replay(False)
    want1 = eager.decode_batch(h1, p1, out_hw=(368, 368)).packed
    want2 = eager.decode_batch(h2, p2, out_hw=(361, 368)).packed
    # rows 361..367 hold peaks in K1 that K2's tables give zero weight: a stale table shows
    assert len(want1['peaks']) > 0 and len(want2['peaks']) > 0
    eng = DecodeEngine(max_batch=5, max_h=368, max_w=368, mode='D')

    def k1():
        eng.enqueue_device(d1[0].data_ptr(), d1[1].data_ptr(), 2, 46, 46, (368, 368))
        return eng.fetch().packed

    def k2():
        eng.enqueue_device(d2[0].data_ptr(), d2[1].data_ptr(), 5, 46, 46, (361, 368))
        return eng.fetch().packed

    for rnd in range(2):
        for it in range(4):
            got = k1()
            for key in KEYS:
                assert np.array_equal(got[key], want1[key]), (rnd, it, key)
        for it in range(1 + 3 * rnd):       # once eagerly; in the second round often enough to be recorded itself
            got = k2()
            for key in KEYS:
                assert np.array_equal(got[key], want2[key]), (rnd, it, key)
    got = k1()
    for key in KEYS:
        assert np.array_equal(got[key], want1[key]), key
    eng.close()
    eager.close()


def test_graph_key_sees_the_skeleton_capacity():
    """The same skeleton pointer with a smaller capacity must not replay the sequence that was
    recorded for the larger one (the pack kernel checks the capacity it was launched with)."""
    import os
    import torch
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    heat, paf = synth.batch(4, H, W, (1, 4), seed0=70, paf_order='xy')
    dh, dp = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
    eng = DecodeEngine(max_batch=4, max_h=H, max_w=W, mode='D')
    buf = torch.full((1 << 16,), 255, dtype=torch.uint8, device='cuda')
    eng.ctx.set_skeleton_buffer(buf.data_ptr(), buf.numel())
    eng.ctx.set_graph_replay(True)      # a context with a skeleton buffer launches eagerly unless asked
    for _ in range(4):
        eng.enqueue_device(dh.data_ptr(), dp.data_ptr(), 4, 23, 23, (H, W))
        eng.fetch()
    small = int(_ffi.res_layout(4)[4]) + 64
    buf.fill_(255)
    eng.ctx.set_skeleton_buffer(buf.data_ptr(), small)
    eng.ctx.set_graph_replay(True)
    eng.enqueue_device(dh.data_ptr(), dp.data_ptr(), 4, 23, 23, (H, W))
    eng.fetch()
    raw = buf.cpu().numpy()
    assert int(raw[:64].view(np.int64)[5]) > small      # reported as too small ...
    assert np.all(raw[small:] == 255)                   # ... and never overrun
    eng.ctx.set_skeleton_buffer(None, 0)
    eng.close()


@pytest.mark.parametrize('caps,exc', [(dict(max_peaks_per_part=2), 'overflow'), (dict(max_persons=1), 'overflow')])
def test_skeleton_only_decode_reports_failed_images(caps, exc):
    """With pd_set_host_results(0) nothing is fetched from the context, so a capacity
    overflow must surface where the skeleton buffers are read."""
    import torch
    from poseestimation_amd import _ffi, sharding, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    heat, paf = synth.batch(2, H, W, 4, seed0=33, paf_order='xy')
    dh, dp = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
    full = dict(max_peaks_per_part=64, max_pairs_per_image=4000, max_persons=32)
    full.update(caps)
    eng = DecodeEngine(max_batch=2, max_h=H, max_w=W, mode='D', **full)
    buf = torch.zeros(1 << 16, dtype=torch.uint8, device='cuda')
    eng.ctx.set_skeleton_buffer(buf.data_ptr(), buf.numel())
    eng.ctx.set_host_results(False)
    eng.enqueue_device(dh.data_ptr(), dp.data_ptr(), 2, 23, 23, (H, W))
    eng.ctx.sync()                       # only waits
    with pytest.raises(_ffi.DecodeOverflow):
        sharding.SkeletonGather.results(buf.cpu().numpy()[None, None])
    eng.ctx.set_skeleton_buffer(None, 0)
    eng.close()


def test_device_resident_path_matches_the_oracle():
    """Network outputs handed over as device pointers (predict.py:61 without the host round
    trip), replayed from the recorded launch sequence: every image against the CPU oracle."""
    import torch
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H, W = 200, 312
    heat, paf = synth.batch(3, H, W, (2, 5), seed0=610, paf_order='xy')
    dh, dp = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
    eng = DecodeEngine(max_batch=3, max_h=H, max_w=W, mode='D')
    for it in range(4):                  # the fourth call is a replay
        eng.enqueue_device(dh.data_ptr(), dp.data_ptr(), 3, heat.shape[1], heat.shape[2], (H, W))
        res = eng.fetch()
    for b in range(3):
        peaks, conns, special, subset = _oracle(heat[b], paf[b], H, W)
        gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
        g_conns, g_special = res.connections(b)
        assert g_special == special
        gio.assert_conns_equal(g_conns, conns, RTOL)
        gio.assert_persons_equal(res.subset(b), subset, RTOL)
    eng.close()


def test_pd_fetch_and_pd_packed_device_agree_with_the_host_mirror():
    import torch
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    heat, paf = synth.batch(3, H, W, (1, 4), seed0=90, paf_order='xy')
    eng = DecodeEngine(max_batch=3, max_h=H, max_w=W, mode='D')
    eng.enqueue(heat, paf)
    a = eng.ctx.fetch()
    b = eng.ctx.fetch_via_pd_fetch()
    for key in KEYS + ('status',):
        assert np.array_equal(a[key], b[key]), key
    peaks_p, conns_p, persons_p, part_p, limb_p, pers_p = eng.ctx.packed_device()
    import ctypes as C

    def dev_bytes(ptr, n):
        t = torch.empty(max(n, 1), dtype=torch.uint8, device='cuda')
        hip = C.CDLL('libamdhip64.so')
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        assert hip.hipMemcpy(t.data_ptr(), ptr, n, 3) == 0     # device to device
        return t.cpu().numpy()[:n]
    assert dev_bytes(peaks_p, a['peaks'].nbytes).tobytes() == a['peaks'].tobytes()
    assert dev_bytes(conns_p, a['conns'].nbytes).tobytes() == a['conns'].tobytes()
    assert dev_bytes(persons_p, a['persons'].nbytes).tobytes() == a['persons'].tobytes()
    assert dev_bytes(part_p, 3 * 18 * 4).view(np.int32).reshape(3, 18).tolist() == a['part_count'].tolist()
    assert dev_bytes(pers_p, 12).view(np.int32).tolist() == a['person_count'].tolist()
    eng.close()


def test_two_contexts_on_one_stream_keep_their_results_apart():
    """pd_share_stream: a second context on the first one's stream; decodes are enqueued on both
    before either is fetched (device-resident inputs, so the recorded launch sequences are used
    too), and each fetch returns its own context's results."""
    import torch
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    batches = [synth.batch(3, H, W, (1, 3), seed0=s, paf_order='xy') for s in (910, 920, 930)]
    dev = [(torch.from_numpy(h).cuda(), torch.from_numpy(p).cuda()) for h, p in batches]
    plain = DecodeEngine(3, H, W, mode='D')
    want = [plain.decode_batch(h, p).packed for h, p in batches]
    a = DecodeEngine(3, H, W, mode='D')
    b = DecodeEngine(3, H, W, mode='D', share_stream_with=a)
    assert a.ctx.stream == b.ctx.stream
    for it in range(8):
        ka, kb = it % 3, (it + 1) % 3
        a.enqueue_device(dev[ka][0].data_ptr(), dev[ka][1].data_ptr(), 3, 23, 23, (H, W))
        b.enqueue_device(dev[kb][0].data_ptr(), dev[kb][1].data_ptr(), 3, 23, 23, (H, W))
        got_a = a.fetch().packed
        a.enqueue_device(dev[kb][0].data_ptr(), dev[kb][1].data_ptr(), 3, 23, 23, (H, W))   # queued behind b's decode
        got_b = b.fetch().packed
        got_a2 = a.fetch().packed
        for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
            assert np.array_equal(got_a[key], want[ka][key]), (it, key)
            assert np.array_equal(got_b[key], want[kb][key]), (it, key)
            assert np.array_equal(got_a2[key], want[kb][key]), (it, key)
    with pytest.raises(Exception):
        b.ctx.submit(dev[0][0].data_ptr(), dev[0][1].data_ptr(), 3, 23, 23, H, W, 0.1, 0.05, 0, 0)
    b.close()
    a.close()
    plain.close()


def test_sharing_context_waits_for_work_queued_outside_decode_batch():
    """On a shared stream a context waits for its own last whole decode through an event.  Work
    queued through the staged entry points or the multi-scale path afterwards must not be
    fetched on the strength of that old event: every fetch returns the results of the work that
    was queued last.  The owner may also be closed first: its stream lives as long as the sharer."""
    import torch
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 368
    B = 16
    big = synth.batch(B, H, W, (4, 8), seed0=4100, paf_order='xy')
    small = synth.batch(B, H, W, (1, 2), seed0=4200, paf_order='xy')
    plain = DecodeEngine(B, H, W, mode='D', fullres=True)
    want_big = plain.decode_batch(*big).packed
    want_small = plain.decode_batch(*small).packed
    want_ms = plain.decode_multiscale([small[0]], [small[1]], [(0, 0)], (H, W)).packed
    a = DecodeEngine(B, H, W, mode='D', fullres=True)
    b = DecodeEngine(B, H, W, mode='D', fullres=True, share_stream_with=a)
    d_big = [torch.from_numpy(x).cuda() for x in big]
    keys = ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons')
    for who in (b, a):      # the sharer and the owner wait the same way
        for it in range(3):
            who.enqueue_device(d_big[0].data_ptr(), d_big[1].data_ptr(), B, 46, 46, (H, W))
            got = who.fetch().packed
            for key in keys:
                assert np.array_equal(got[key], want_big[key]), (it, key)
            # staged entry points on the same context, other input
            who.ctx.set_lowres(small[0].ctypes.data, small[1].ctypes.data, B, 46, 46, H, W, _ffi.MEM_HOST)
            who.ctx.peaks(0.1)
            who.ctx.connections(0.05)
            who.ctx.parse()
            who.ctx.pack()
            got = who.ctx.fetch()
            for key in keys:
                assert np.array_equal(got[key], want_small[key]), (it, key)
            who.enqueue_device(d_big[0].data_ptr(), d_big[1].data_ptr(), B, 46, 46, (H, W))
            who.fetch()
            got = who.decode_multiscale([small[0]], [small[1]], [(0, 0)], (H, W)).packed
            for key in keys:
                assert np.array_equal(got[key], want_ms[key]), (it, key)
    a.close()               # the owner first: its stream must outlive b
    b.enqueue_device(d_big[0].data_ptr(), d_big[1].data_ptr(), B, 46, 46, (H, W))
    got = b.fetch().packed
    for key in keys:
        assert np.array_equal(got[key], want_big[key]), key
    b.close()
    plain.close()


def test_person_rows_keep_their_places_through_changing_batches_and_busy_streams():
    """Assembly and packing in one launch place each image's person rows by a look-back over the
    row counts of the images before it, through words that carry the launch's tag.  Batches of
    changing size through one context (fewer images, then more again: the words of the images
    beyond a batch must not be taken for this launch's), replayed sequences included, and four
    contexts on streams of their own decoding at once, many times over: the packed bytes must be
    those of a context that keeps assembly and packing apart."""
    import os
    import subprocess
    import sys
    import torch
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 184
    heat, paf = synth.batch(24, H, W, (0, 6), seed0=900, paf_order='xy')
    heat[5] = 0          # an image without anybody in the middle of the batch
    d_heat, d_paf = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
    # the reference results come from a process with PD_SEPARATE_PACK=1 (the switch is read once per process)
    code = ('import numpy as np, sys\n'
            'from poseestimation_amd import synth\n'
            'from poseestimation_amd.predict_decode import DecodeEngine\n'
            'heat, paf = synth.batch(24, %d, %d, (0, 6), seed0=900, paf_order="xy")\n'
            'heat[5] = 0\n'
            'eng = DecodeEngine(max_batch=24, max_h=%d, max_w=%d, mode="D")\n'
            'out = {}\n'
            'for B in (24, 3, 1, 17, 24):\n'
            '    r = eng.decode_batch(heat[:B], paf[:B]).packed\n'
            '    for k in %r:\n'
            '        out["%%d_%%s" %% (B, k)] = r[k]\n'
            'np.savez(sys.argv[1], **out)\n' % (H, W, H, W, KEYS))
    path = os.path.join(os.environ.get('TMPDIR', '/tmp'), 'pd_separate_pack_%d.npz' % os.getpid())
    env = dict(os.environ, PD_SEPARATE_PACK='1', PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    subprocess.run([sys.executable, '-c', code, path], check=True, env=env, timeout=240)
    want = dict(np.load(path))
    os.remove(path)
    assert want['24_person_count'].sum() > 20 and want['24_person_count'][5] == 0

    def check(res, B, where):
        for k in KEYS:
            assert np.array_equal(res[k], want['%d_%s' % (B, k)]), (where, B, k)

    eng = DecodeEngine(max_batch=24, max_h=H, max_w=W, mode='D')
    for rep in range(3):                        # (the third identical call replays a recorded sequence)
        for B in (24, 3, 1, 17, 24, 24, 24):
            eng.enqueue_device(d_heat.data_ptr(), d_paf.data_ptr(), B, H // 8, W // 8)
            check(eng.fetch().packed, B, 'one context, pass %d' % rep)
    # four contexts at once
    engs = [eng] + [DecodeEngine(max_batch=24, max_h=H, max_w=W, mode='D') for _ in range(3)]
    for it in range(40):
        sizes = [(24, 17, 3, 24)[(it + j) % 4] for j in range(4)]
        for e, B in zip(engs, sizes):
            e.enqueue_device(d_heat.data_ptr(), d_paf.data_ptr(), B, H // 8, W // 8)
        for e, B in zip(engs, sizes):
            check(e.fetch().packed, B, 'four contexts, round %d' % it)
    for e in engs:
        e.close()
