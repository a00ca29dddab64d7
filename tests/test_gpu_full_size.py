"""The benchmark configurations at their full sizes: every image of the 64 x 368^2
batch against the CPU oracle; the 256 x 656^2 / 30-people batch through a sample of
images against the oracle plus size-independent properties on all of them; the
four-scale configuration at a reduced batch."""
import numpy as np
import pytest

from oracle import decode_ref as ref
from tests import golden_io as gio

pytestmark = pytest.mark.gpu

RTOL = 1e-5


def _oracle_image(heat, paf, mode, H, W):
    hf = ref.resize_cubic(heat, fx=8, fy=8)[:H, :W].astype(np.float64)
    pf = ref.resize_cubic(paf, fx=8, fy=8)[:H, :W].astype(np.float64)
    peaks = ref.get_peaks(hf, 0.1, mode, use_scipy=True)
    conns, special = ref.get_connections(peaks, pf, 0.05, mode)
    subset = ref.multi_person_parse(peaks, conns, special, mode)
    return peaks, conns, special, subset


def _assert_image(res, b, want, mode):
    peaks, conns, special, subset = want
    gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
    g_conns, g_special = res.connections(b)
    assert g_special == special
    gio.assert_conns_equal(g_conns, conns, RTOL)
    gio.assert_persons_equal(res.subset(b), subset, RTOL)


@pytest.mark.parametrize('mode', ['D', 'P'])
def test_batch_64_at_368_every_image_matches_the_oracle(mode):
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 64, 368, 368
    heat, paf = synth.batch(B, H, W, (1, 8), seed0=0, paf_order='xy' if mode == 'D' else 'yx')
    eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode=mode, max_peaks_per_part=128, max_persons=128,
                       max_pairs_per_image=40000)
    res = eng.decode_batch(heat, paf)
    stats = eng.ctx.filter_stats()
    assert stats['planes_exact'] == 0 and stats['candidates'] >= res.packed['peaks'].shape[0]
    # a peak stage that visits a part of the map: some cells of both kinds, far from all of them
    assert 0 < stats['cells_border'] < stats['cells_interior'] < B * 19 * 23 * 23 // 2
    again = eng.decode_batch(heat, paf)
    for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        assert np.array_equal(res.packed[key], again.packed[key]), key       # deterministic
    for b in range(B):
        _assert_image(res, b, _oracle_image(heat[b], paf[b], mode, H, W), mode)
    # an image decodes the same alone, and wherever it sits in the batch
    perm = np.random.RandomState(1).permutation(B)
    shuffled = eng.decode_batch(heat[perm], paf[perm])
    solo = DecodeEngine(max_batch=1, max_h=H, max_w=W, mode=mode)
    for i in (0, 17, 63):
        b = int(perm[i])
        np.testing.assert_array_equal(shuffled.subset(i), res.subset(b))
        assert shuffled.all_peaks(i) == res.all_peaks(b)
        one = solo.decode_batch(heat[b:b + 1], paf[b:b + 1])
        np.testing.assert_array_equal(one.subset(0), res.subset(b))
    solo.close()
    eng.close()


def test_batch_256_at_656_with_30_people():
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 256, 656, 656
    heat, paf = synth.batch(B, H, W, 30, seed0=0, paf_order='xy')
    eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                       max_pairs_per_image=40000)
    res = eng.decode_batch(heat, paf)
    stats = eng.ctx.filter_stats()
    assert stats['planes_exact'] == 0 and stats['candidates'] > 0
    # full oracle check on a sample (about 2 s of CPU per image)
    for b in (0, 101, 255):
        _assert_image(res, b, _oracle_image(heat[b], paf[b], 'D', H, W), 'D')
    n_persons = 0
    for b in range(B):
        peaks = res.all_peaks(b)
        conns, special = res.connections(b)
        subset = res.subset(b)
        n_persons += len(subset)
        # person assembly re-derived on the CPU from the GPU's own peaks and connections
        want = ref.multi_person_parse(peaks, conns, special, 'D')
        gio.assert_persons_equal(subset, want, RTOL)
        # peak lists: row-major order inside a part, ids consecutive over the image
        flat = [t for pl in peaks for t in pl]
        assert [t[3] for t in flat] == list(range(len(flat)))
        for pl in peaks:
            keys = [int(t[1]) * W + int(t[0]) for t in pl]      # (x, y, score, id) tuples
            assert keys == sorted(keys) and len(set(keys)) == len(keys)
        # connections: descending score, every peak used at most once per limb side
        for k, c in enumerate(conns):
            if k in special:
                continue
            assert np.all(np.diff(c[:, 2]) <= 0)
            assert len(set(c[:, 0])) == len(c) and len(set(c[:, 1])) == len(c)
            assert len(c) <= min(len(peaks[ref.LIMB_FROM[k]]), len(peaks[ref.LIMB_TO[k]]))
        # persons: pruning rule; the part counter never undercounts (it overcounts when a
        # part is overwritten, predict.py:222-225)
        for row in subset:
            assert row[19] >= 4 and row[18] / row[19] >= 0.4
            assert row[19] >= np.sum(row[:18] >= 0)
    assert n_persons > 20 * B        # about 30 people per image, most of them recovered
    eng.close()


def test_four_scales_at_368():
    """predict.py:49-72 with scale_search (0.5, 1, 1.5, 2) on 368^2 images: network outputs
    of 23^2, 46^2, 69^2 and 92^2, every scale resized to 368^2 and averaged, then decoded."""
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 6, 368, 368
    scales = [0.5, 1.0, 1.5, 2.0]
    heats, pafs = [], []
    for s in scales:
        outs = [synth.scene(H, W, 1 + b % 5, 900 + b, paf_order='xy', scale=s) for b in range(B)]
        heats.append(np.stack([o[0] for o in outs]))
        pafs.append(np.stack([o[1] for o in outs]))
    assert [h.shape[1] for h in heats] == [23, 46, 69, 92]
    pads = [(0, 0)] * 4
    eng = DecodeEngine(B, H, W, mode='D')
    res = eng.decode_multiscale(heats, pafs, pads, (H, W))
    for b in (0, 3, 5):
        want_h, want_p = ref.upsample_average([h[b] for h in heats], [p[b] for p in pafs], pads, (H, W))
        cand, peaks, subset = ref.decode(want_h, want_p, 0.1, 0.05, 'D')
        gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
        np.testing.assert_array_equal(res.candidate(b), cand)
        gio.assert_persons_equal(res.subset(b), subset, RTOL)
    eng.close()


def test_multiscale_skeletons_leave_through_the_gather_buffer():
    """The multi-scale decode of a sharded job: with a skeleton buffer set, the decode of the
    averaged maps writes peaks and person rows there (host results on, then off), and a round
    of SkeletonGather(local=True) returns them: equal to the fetched results both times."""
    from poseestimation_amd import sharding, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 5, 184, 184
    scales = [0.5, 1.0, 2.0]
    heats, pafs = [], []
    for s in scales:
        outs = [synth.scene(H, W, 1 + b % 4, 1200 + b, paf_order='xy', scale=s) for b in range(B)]
        heats.append(np.stack([o[0] for o in outs]))
        pafs.append(np.stack([o[1] for o in outs]))
    pads = [(0, 0)] * 3
    eng = DecodeEngine(B, H, W, mode='D')
    want = eng.decode_multiscale(heats, pafs, pads, (H, W)).packed
    assert want['person_count'].sum() > 0
    g = sharding.SkeletonGather([eng.ctx], 1 << 16, depth=1, local=True)
    for host_results in (True, False):
        eng.ctx.set_host_results(host_results)
        for s in (0, 1):
            g.point(s, 0, 0)
            if host_results:
                got_host = eng.decode_multiscale(heats, pafs, pads, (H, W)).packed
                assert np.array_equal(got_host['persons'], want['persons'])
            else:
                eng.upsample_average(heats, pafs, pads, (H, W))
                eng.ctx.decode_fullres(0.1, 0.05)
            g.start(s)
        for s in (0, 1):
            got = sharding.SkeletonGather.results(g.finish(s))[0]
            for key in ('part_count', 'person_count', 'status', 'peaks', 'persons'):
                assert np.array_equal(got[key], want[key]), (host_results, s, key)
    eng.ctx.set_host_results(True)
    eng.ctx.set_skeleton_buffer(None, 0)
    eng.close()


def test_config4_batch_128_four_scales():
    """BASELINE config 4 at its size on one GPU: 128 images x scales (0.5, 1, 1.5, 2) at 368^2
    through predict.py:49-72 (x8 bicubic, resize, average) and the decode.  A sample of images
    against the oracle's upsample_average + decode, size-independent properties on all 128."""
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    B, H, W = 128, 368, 368
    scales = [0.5, 1.0, 1.5, 2.0]
    heats, pafs = [], []
    for s in scales:
        outs = [synth.scene(H, W, 1 + b % 8, 4000 + b, paf_order='xy', scale=s) for b in range(B)]
        heats.append(np.stack([o[0] for o in outs]))
        pafs.append(np.stack([o[1] for o in outs]))
    assert [h.shape[1] for h in heats] == [23, 46, 69, 92]
    pads = [(0, 0)] * 4
    eng = DecodeEngine(B, H, W, mode='D', max_peaks_per_part=128, max_persons=128, max_pairs_per_image=40000)
    res = eng.decode_multiscale(heats, pafs, pads, (H, W))
    again = eng.decode_multiscale(heats, pafs, pads, (H, W))
    for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        assert np.array_equal(res.packed[key], again.packed[key]), key       # deterministic
    # the form that writes the averaged maps out and decodes those: the same bytes
    stored = eng.decode_multiscale(heats, pafs, pads, (H, W), materialise=True)
    for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        assert np.array_equal(res.packed[key], stored.packed[key]), key
    for b in (0, 77, 127):
        want_h, want_p = ref.upsample_average([h[b] for h in heats], [p[b] for p in pafs], pads, (H, W))
        cand, peaks, subset = ref.decode(want_h, want_p, 0.1, 0.05, 'D')
        gio.assert_peaks_equal(res.all_peaks(b), peaks, exact_score=True)
        np.testing.assert_array_equal(res.candidate(b), cand)
        conns, special = ref.get_connections(peaks, want_p, 0.05, 'D')
        g_conns, g_special = res.connections(b)
        assert g_special == special
        gio.assert_conns_equal(g_conns, conns, RTOL)
        gio.assert_persons_equal(res.subset(b), subset, RTOL)
    n_persons = 0
    for b in range(B):
        peaks = res.all_peaks(b)
        conns, special = res.connections(b)
        subset = res.subset(b)
        n_persons += len(subset)
        gio.assert_persons_equal(subset, ref.multi_person_parse(peaks, conns, special, 'D'), RTOL)
        flat = [t for pl in peaks for t in pl]
        assert [t[3] for t in flat] == list(range(len(flat)))
        for pl in peaks:
            keys = [int(t[1]) * W + int(t[0]) for t in pl]
            assert keys == sorted(keys) and len(set(keys)) == len(keys)
        for k, c in enumerate(conns):
            if k in special:
                continue
            assert np.all(np.diff(c[:, 2]) <= 0)
            assert len(set(c[:, 0])) == len(c) and len(set(c[:, 1])) == len(c)
    assert n_persons >= B        # 1..8 people per image
    # an image decodes the same alone as inside the batch
    solo = DecodeEngine(1, H, W, mode='D')
    for b in (5, 100):
        one = solo.decode_multiscale([h[b:b + 1] for h in heats], [p[b:b + 1] for p in pafs], pads, (H, W))
        np.testing.assert_array_equal(one.subset(0), res.subset(b))
        assert one.all_peaks(0) == res.all_peaks(b)
    solo.close()
    eng.close()


def test_config5_1024_images_streamed_in_chunks():
    """BASELINE config 5 on one GPU: 1024 images of 46x46x57 streamed in chunks of 64 through
    the round pipeline (four skeleton-only contexts, launcher threads, two rounds in flight,
    the gather replaced by the copy to pinned memory).  Every chunk equals a plain decode of
    the same chunk, image order is restored, a sample is checked against the oracle."""
    import torch
    from poseestimation_amd import sharding, synth
    from poseestimation_amd.predict_decode import DecodeEngine, Results
    H = W = 368
    CH, NCH = 64, 16
    base_h, base_p = synth.batch(256, H, W, (1, 8), seed0=9000, paf_order='xy')
    # 1024 images: the 256 distinct ones in four different rotations
    heat = np.concatenate([np.roll(base_h, -37 * r, axis=0) for r in range(4)])
    paf = np.concatenate([np.roll(base_p, -37 * r, axis=0) for r in range(4)])
    d_heat, d_paf = torch.from_numpy(heat).cuda(), torch.from_numpy(paf).cuda()
    plain = DecodeEngine(max_batch=CH, max_h=H, max_w=W, mode='D', max_pairs_per_image=40000)
    want = [plain.decode_batch(heat[c * CH:(c + 1) * CH], paf[c * CH:(c + 1) * CH]).packed for c in range(NCH)]
    plain.close()
    engs = [DecodeEngine(max_batch=CH, max_h=H, max_w=W, mode='D', max_pairs_per_image=40000) for _ in range(4)]
    for e in engs:
        e.ctx.set_host_results(False)
    cap = 1 << 18
    g = sharding.SkeletonGather([e.ctx for e in engs], cap, depth=2, local=True)
    hb = CH * 46 * 46 * 19 * 4
    pb = CH * 46 * 46 * 38 * 4
    got = {}
    for rep in range(3):                 # the third pass replays recorded launch sequences
        pending = [None, None]
        for r in range(NCH // g.n + 2):
            s = r % 2
            if pending[s] is not None:
                for j, res in enumerate(sharding.SkeletonGather.results(g.finish(s))):
                    got[pending[s][j]] = res
                pending[s] = None
            chunks = list(range(r * g.n, min(NCH, (r + 1) * g.n)))
            if not chunks:
                continue
            for j, c in enumerate(chunks):
                engs[j % 4].ctx.submit(d_heat.data_ptr() + c * hb, d_paf.data_ptr() + c * pb, CH, 46, 46, H, W,
                                       0.1, 0.05, g.slice_ptr(s, j), cap)
            for e in engs:
                e.ctx.flush()
            g.start(s)
            pending[s] = chunks
        assert sorted(got) == list(range(NCH))
        for c in range(NCH):
            for key in ('part_count', 'person_count', 'status', 'peaks', 'persons'):
                assert np.array_equal(got[c][key], want[c][key]), (rep, c, key)
    # image order: the chunks concatenated in order are the 1024 images in order
    allres = sharding.concat_results([got[c] for c in range(NCH)])
    assert allres['B'] == 1024
    r = Results(allres, engs[0].mode)
    for b in (0, 300, 1023):
        peaks, conns, special, subset = _oracle_image(heat[b], paf[b], 'D', H, W)
        gio.assert_peaks_equal(r.all_peaks(b), peaks, exact_score=True)
        gio.assert_persons_equal(r.subset(b), subset, RTOL)
    # rotation r of the base batch sits at images 256 r ..: image 256 r + i is base image (i + 37 r) % 256
    for rr in (1, 3):
        for i in (0, 200):
            np.testing.assert_array_equal(r.subset(256 * rr + i)[:, 18:], r.subset((i + 37 * rr) % 256)[:, 18:])
    for e in engs:
        e.close()


def test_batch_beyond_one_round_of_person_placement():
    """Person assembly and packing share a launch whose last workgroup places the person rows of
    all images, 256 images per round: 600 small images (some empty) in one call against the same
    images decoded in batches of 100, and a sample against the oracle."""
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 96
    B = 600
    heat, paf = synth.batch(B, H, W, (0, 3), seed0=7000, paf_order='xy')
    eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=32, max_persons=32,
                       max_pairs_per_image=2000)
    res = eng.decode_batch(heat, paf)
    small = DecodeEngine(max_batch=100, max_h=H, max_w=W, mode='D', max_peaks_per_part=32, max_persons=32,
                         max_pairs_per_image=2000)
    n_persons = 0
    for b0 in range(0, B, 100):
        part = small.decode_batch(heat[b0:b0 + 100], paf[b0:b0 + 100])
        for i in range(100):
            gio.assert_peaks_equal(res.all_peaks(b0 + i), part.all_peaks(i), exact_score=True)
            np.testing.assert_array_equal(res.subset(b0 + i), part.subset(i))
            n_persons += len(part.subset(i))
    assert n_persons > 300 and len(res.packed['persons']) == n_persons
    for b in (0, 255, 256, 257, 511, 512, 599):
        _assert_image(res, b, _oracle_image(heat[b], paf[b], 'D', H, W), 'D')
    eng.close()
    small.close()


def test_more_than_64_open_persons_inside_the_fused_decode():
    """An image with more open persons than the one-lane-per-person form holds is handed to the
    table form inside the same launch (one wavefront of the image's workgroup runs it)."""
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 656
    heat = np.empty((2, H // 8, W // 8, 19), np.float32)
    paf = np.empty((2, H // 8, W // 8, 38), np.float32)
    heat[0], paf[0] = synth.scene(H, W, 90, 901, paf_order='xy')
    heat[1], paf[1] = synth.scene(H, W, 5, 902, paf_order='xy')
    eng = DecodeEngine(max_batch=2, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=200,
                       max_pairs_per_image=200000)
    res = eng.decode_batch(heat, paf)
    want = _oracle_image(heat[0], paf[0], 'D', H, W)
    assert len(want[3]) > 64
    for b in range(2):
        _assert_image(res, b, _oracle_image(heat[b], paf[b], 'D', H, W), 'D')
    eng.close()


def test_pair_scoring_with_and_without_its_pretest_agree():
    """The pair kernel comes with and without the pre-test of samples 4, 5, 2 and 7; a context picks
    by the peaks per image of its last results.  A crowded batch right after a sparse one is scored
    without the pre-test, the same batch again with it: both equal the oracle."""
    from poseestimation_amd import synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 368
    sparse_h, sparse_p = synth.batch(4, H, W, 1, seed0=8100, paf_order='xy')
    crowd_h, crowd_p = synth.batch(4, H, W, 16, seed0=8200, paf_order='xy')
    eng = DecodeEngine(max_batch=4, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                       max_pairs_per_image=40000)
    eng.decode_batch(sparse_h, sparse_p)
    first = eng.decode_batch(crowd_h, crowd_p)      # chosen by the sparse results: no pre-test
    second = eng.decode_batch(crowd_h, crowd_p)     # chosen by the crowded results: pre-test
    for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        assert np.array_equal(first.packed[key], second.packed[key]), key
    assert len(first.packed['peaks']) / 4 > 160       # the rule's limit, so that the two decodes did differ in kernel
    for b in range(4):
        _assert_image(first, b, _oracle_image(crowd_h[b], crowd_p[b], 'D', H, W), 'D')
    eng.close()


@pytest.mark.parametrize('cap', [4, 64])
def test_matching_beyond_the_sort_list_takes_the_arg_max_form(cap):
    """A limb with more surviving pairs than the wavefront's sort list holds is matched by repeated
    arg-max instead: the same connections.  The list is shrunk to force it (crowded scene, both
    modes' tie rules covered by the oracle comparison)."""
    import ctypes as C
    from poseestimation_amd import _ffi, synth
    from poseestimation_amd.predict_decode import DecodeEngine
    H = W = 368
    heat, paf = synth.batch(3, H, W, 12, seed0=8300, paf_order='xy')
    lib = C.CDLL(_ffi.LIB_PATH)
    knob = C.c_int.in_dll(lib, 'pd_match_cap_override')
    eng = DecodeEngine(max_batch=3, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                       max_pairs_per_image=40000)
    knob.value = cap
    try:
        res = eng.decode_batch(heat, paf)
    finally:
        knob.value = 0
    n_conn = 0
    for b in range(3):
        want = _oracle_image(heat[b], paf[b], 'D', H, W)
        n_conn += max(len(c) for c in want[1])
        _assert_image(res, b, want, 'D')
    assert n_conn > 3 * cap or cap > 4      # with cap 4 every busy limb overflows the list
    eng.close()


_SWITCH_SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from poseestimation_amd import synth
from poseestimation_amd.predict_decode import DecodeEngine
out = hashlib.sha256()
stats = hashlib.sha256()
for (B, H, W, people) in ((6, 368, 368, (1, 8)), (3, 236, 323, (1, 4)), (2, 656, 656, 12)):
    heat, paf = synth.batch(B, H, W, people, seed0=4200, paf_order='xy')
    eng = DecodeEngine(max_batch=B, max_h=H, max_w=W, mode='D', max_peaks_per_part=128, max_persons=128,
                       max_pairs_per_image=40000)
    for _ in range(4):      # eager, recording, replays
        res = eng.decode_batch(heat, paf, out_hw=(H, W))
    for key in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        out.update(np.ascontiguousarray(res.packed[key]).tobytes())
    st = eng.ctx.filter_stats()
    stats.update(repr(sorted(st.items())).encode())
    eng.close()
print('DIGEST', out.hexdigest(), stats.hexdigest())
'''


def test_comparison_switches_leave_the_results_alone():
    """The environment switches that keep kernels apart (or put the layout kernel in front) are
    read once per process: a child process per setting decodes the same batches; packed results
    and filter statistics must be the same bytes as with no switch set.  (PD_TILE_FILTER, the
    tile form of the small-map filter, takes its tolerance from the maximum of a tile where the
    per-plane form takes it from the whole plane: the same results, not the same candidate counts.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    digests = {}
    for name in ('', 'PD_TILE_FILTER', 'PD_SPLIT_CELLS', 'PD_PLANAR_FIRST', 'PD_PLANAR_COPY', 'PD_SEPARATE_PACK', 'PD_SEPARATE_SORT'):
        env = {k: v for k, v in os.environ.items() if not k.startswith('PD_')}
        if name:
            env[name] = '1'
        p = subprocess.run([sys.executable, '-c', _SWITCH_SCRIPT % root], env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=100)
        assert p.returncode == 0, p.stdout[-2000:]
        lines = [l for l in p.stdout.splitlines() if l.startswith('DIGEST ')]
        assert len(lines) == 1, p.stdout[-2000:]
        digests[name or 'default'] = lines[0].split()[1:]
    assert len(set(d[0] for d in digests.values())) == 1, digests
    # the switches that only regroup launches see the same cells and candidates
    assert len(set(d[1] for k, d in digests.items() if k != 'PD_TILE_FILTER')) == 1, digests
