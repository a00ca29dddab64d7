"""CPU-side tests: the C-ABI library loads and exports what include/posedecode.h
declares (no compute without a GPU), the host logic, the result blobs and the
world_size-2 gather over gloo."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    text = open(os.path.join(ROOT, 'include', 'posedecode.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pd_[a-z_0-9]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from poseestimation_amd import _ffi
    lib = _ffi.lib()
    declared = _header_functions()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), 'libposedecode.so does not export %s' % name
    assert sorted(_ffi.EXPORTS) == declared
    assert b'gfx950' in lib.pd_version()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    from poseestimation_amd import _ffi
    with pytest.raises(RuntimeError):
        _ffi.Context(max_batch=1, max_h=64, max_w=64)
    from poseestimation_amd import postprocess
    with pytest.raises(RuntimeError):
        postprocess.get_peaks(np.zeros((40, 40, 18)), 0.1)


def test_product_package_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, 'poseestimation_amd')
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            src = open(os.path.join(pkg, fn)).read()
            assert 'oracle' not in src.replace('oracle/decode_ref.py', ''), fn


def test_config_tables_match_reference_values():
    from poseestimation_amd import config
    assert config.limb_seq[0] == [1, 8] and config.limb_seq[18] == [15, 17]
    assert config.map_idx[7] == [14, 15]
    assert len(config.limb_seq) == 19 and len(config.coco_to_ours) == 17
    from oracle import decode_ref as ref
    assert tuple(ref.LIMB_FROM) == tuple(config.LIMB_FROM) and tuple(ref.LIMB_TO) == tuple(config.LIMB_TO)


def test_pad_right_down_corner():
    from poseestimation_amd.predict_decode import pad_right_down_corner
    img = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    out, pad = pad_right_down_corner(img, 8, 128)
    assert pad == [0, 0, 3, 1] and out.shape == (8, 8, 3)
    assert np.array_equal(out[:5, :7], img) and np.all(out[5:] == 128) and np.all(out[:, 7:] == 128)
    out, pad = pad_right_down_corner(np.zeros((16, 8, 3), np.uint8), 8, 128)
    assert pad == [0, 0, 0, 0] and out.shape == (16, 8, 3)


def test_helpers_is_isolated_and_id_to_idx():
    from poseestimation_amd.postprocess import _id_to_idx, is_isolated
    a = -np.ones(18); b = -np.ones(18)
    a[3] = 5; b[4] = 6
    assert is_isolated(a, b)
    b[3] = 0
    assert not is_isolated(a, b)
    peaks = [(1, 2, 0.5, 10), (3, 4, 0.6, 11), (5, 6, 0.7, 11)]
    assert _id_to_idx(peaks, 10) == 0 and _id_to_idx(peaks, 11) == 2
    with pytest.raises(UnboundLocalError):
        _id_to_idx(peaks, 99)


def test_shard_range_covers_batch_in_order():
    from poseestimation_amd.sharding import shard_range
    for n in (1, 7, 64, 128, 1000, 1024):
        for world in (1, 2, 4, 8):
            spans = [shard_range(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a0 <= a1
            assert max(hi - lo for lo, hi in spans) == -(-n // world)


def _fake_results(B, seed):
    from poseestimation_amd._ffi import CONN_DTYPE, PEAK_DTYPE
    rng = np.random.RandomState(seed)
    part = rng.randint(0, 4, (B, 18)).astype(np.int32)
    limb = rng.randint(-1, 3, (B, 19)).astype(np.int32)
    pers = rng.randint(0, 3, (B,)).astype(np.int32)
    pk = np.zeros(int(part.sum()), PEAK_DTYPE)
    pk['y'] = rng.randint(0, 368, len(pk)); pk['x'] = rng.randint(0, 368, len(pk)); pk['score'] = rng.rand(len(pk))
    cn = np.zeros(int(np.maximum(limb, 0).sum()), CONN_DTYPE)
    cn['id_a'] = rng.randint(0, 50, len(cn)); cn['score'] = rng.rand(len(cn))
    return {'B': B, 'part_count': part, 'limb_count': limb, 'person_count': pers,
            'status': np.zeros(B, np.int32), 'peaks': pk, 'conns': cn,
            'persons': rng.rand(int(pers.sum()), 20)}


def test_result_blob_round_trip_and_concat():
    from poseestimation_amd import sharding
    r0, r1 = _fake_results(5, 1), _fake_results(3, 2)
    b0 = sharding.pack_blob(r0)
    assert sharding.blob_nbytes(np.concatenate([b0, np.zeros(100, np.uint8)])) == b0.size
    u0 = sharding.unpack_blob(b0)
    for k in r0:
        assert np.array_equal(u0[k], r0[k]), k
    merged = sharding.concat_results([u0, sharding.unpack_blob(sharding.pack_blob(r1))])
    assert merged['B'] == 8 and len(merged['peaks']) == len(r0['peaks']) + len(r1['peaks'])
    assert np.array_equal(merged['part_count'][5:], r1['part_count'])


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch.distributed as dist
from poseestimation_amd import sharding
from tests.test_host import _fake_results
dist.init_process_group('gloo', init_method='tcp://127.0.0.1:%(port)d', rank=int(sys.argv[1]), world_size=2)
rank = dist.get_rank()
lo, hi = sharding.shard_range(9, 2, rank)
res = _fake_results(hi - lo, 40 + rank)
out = sharding.gather_results(res, 1 << 16)
if rank == 0:
    want = sharding.concat_results([_fake_results(5, 40), _fake_results(4, 41)])
    assert out['B'] == 9
    for k in ('part_count', 'limb_count', 'person_count', 'peaks', 'conns', 'persons'):
        assert np.array_equal(out[k], want[k]), k
    print('GATHER_OK')
else:
    assert out is None

# the exchange of the bench: skeleton buffers (no connection lists), one gather per round of
# two contexts, two rounds in flight
import torch
class FakeCtx:
    def set_skeleton_buffer(self, ptr, cap):
        self.ptr, self.cap = ptr, cap
ctxs = [FakeCtx(), FakeCtx()]
cap = 1 << 15
g = sharding.SkeletonGather(ctxs, cap, device=torch.device('cpu'))
def skeleton(n, seed):
    r = _fake_results(n, seed)
    r['conns'] = r['conns'][:0]
    r['limb_count'] = np.zeros_like(r['limb_count'])
    return r
for s in (0, 1):
    for i in range(2):
        g.point(s, i, i)
    assert [c.ptr for c in ctxs] == [g.send[s].data_ptr(), g.send[s].data_ptr() + cap] and ctxs[0].cap == cap
    for i in range(2):          # what the pack kernels of the two contexts would write
        blob = sharding.pack_blob(skeleton(hi - lo, 70 + 10 * s + 3 * i + rank))
        g.send[s][i * cap:i * cap + blob.size] = torch.from_numpy(blob)
    g.start(s)
for s in (0, 1):
    rows = g.finish(s)
    if rank == 0:
        assert rows.shape == (2, 2, cap)
        got = sharding.SkeletonGather.results(rows)
        sizes = [sharding.shard_range(9, 2, r) for r in range(2)]
        for i in range(2):
            want = sharding.concat_results([skeleton(b - a, 70 + 10 * s + 3 * i + r) for r, (a, b) in enumerate(sizes)])
            assert got[i]['B'] == 9 and len(got[i]['conns']) == 0
            for k in ('part_count', 'person_count', 'peaks', 'persons'):
                assert np.array_equal(got[i][k], want[k]), k
    else:
        assert rows is None
if rank == 0:
    bad = g.recv[0].numpy().reshape(2, 2, cap).copy()
    bad[1, 0, :64].view(np.int64)[5] = 123456
    try:
        sharding.SkeletonGather.results(bad)
        raise SystemExit('a truncated blob was accepted')
    except ValueError:
        pass
    print('SKELETON_GATHER_OK')
dist.barrier()
dist.destroy_process_group()
'''


def test_gather_world_size_2_gloo(tmp_path):
    """The N > 1 path: two ranks decode their contiguous blocks (results faked on
    CPU), one fixed-size gather to rank 0 restores image order."""
    import socket
    s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT, 'port': port})
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert 'GATHER_OK' in outs[0] and 'SKELETON_GATHER_OK' in outs[0]


def test_synth_is_deterministic_and_shaped():
    from poseestimation_amd import synth
    h1, p1 = synth.scene(368, 368, 3, 7)
    h2, p2 = synth.scene(368, 368, 3, 7)
    assert h1.shape == (46, 46, 19) and p1.shape == (46, 46, 38) and h1.dtype == np.float32
    assert np.array_equal(h1, h2) and np.array_equal(p1, p2)
    hb, pb = synth.batch(3, 184, 248, (1, 4), seed0=5)
    assert hb.shape == (3, 23, 31, 19) and pb.shape == (3, 23, 31, 38)


def test_coco_records_match_oracle_restatement():
    """evaluate.py:44-74 (host-side consumer of the decode)."""
    from oracle import decode_ref as ref
    from poseestimation_amd import coco, config
    rng = np.random.RandomState(3)
    cand = np.concatenate([rng.randint(0, 368, (40, 2)).astype(float), rng.rand(40, 1),
                           np.arange(40.)[:, None]], axis=1)
    subset = -np.ones((4, 20))
    for p in range(4):
        idx = rng.permutation(18)[:rng.randint(4, 18)]
        subset[p, idx] = rng.randint(0, 40, len(idx))
        subset[p, 18] = rng.rand() * 20
        subset[p, 19] = len(idx)
    got = coco.coco_records(cand, subset, 42)
    want = ref.coco_records(cand, subset, 42, config.coco_to_ours)
    assert len(got) == len(want) == 4
    for g, w in zip(got, want):
        assert g['image_id'] == 42 and g['category_id'] == 1 and g['score'] == w['score']
        assert [float(v) for v in g['keypoints']] == [float(v) for v in w['keypoints']]
    assert coco.coco_records(np.zeros((0, 4)), np.zeros((0, 20)), 1) == []


def test_image_resize_shape_through_the_c_abi():
    """pd_image_resize_shape is host arithmetic (no device work): cv2.resize's destination size
    for fx = fy = scale is the product rounded half to even, the pad is predict.py:23-24."""
    import ctypes as C
    from poseestimation_amd import _ffi
    L = _ffi.lib()
    for h, w, scale, stride in [(97, 131, 0.5 * 368 / 97, 8), (480, 640, 368 / 480.0, 8), (5, 5, 0.5, 8), (64, 48, 1.0, 8),
                                (33, 35, 4.1, 4)]:
        v = [C.c_int32() for _ in range(4)]
        assert L.pd_image_resize_shape(h, w, scale, stride, *[C.byref(x) for x in v]) == 0
        rh, rw = int(round(h * scale)), int(round(w * scale))
        assert [x.value for x in v] == [rh, rw, (stride - rh % stride) % stride, (stride - rw % stride) % stride]
    assert L.pd_image_resize_shape(0, 5, 1.0, 8, None, None, None, None) == _ffi.PD_E_INVALID
    assert L.pd_image_resize_shape(5, 5, 0.0, 8, None, None, None, None) == _ffi.PD_E_INVALID


def test_oracle_image_resize_pad_properties():
    from oracle import decode_ref as ref
    rng = np.random.RandomState(4)
    img = rng.randint(0, 256, size=(40, 52, 3)).astype(np.uint8)
    same, pad = ref.image_resize_pad(img, 1.0, 8, 128)
    assert pad == [0, 0, 0, 4] and same.dtype == np.uint8
    assert np.array_equal(same[:, :52], img) and np.all(same[:, 52:] == 128)
    flat = np.full((30, 30, 3), 77, np.uint8)
    up, pad = ref.image_resize_pad(flat, 2.3, 8, 128)
    assert up.shape == (72, 72, 3) and pad == [0, 0, 3, 3]
    assert np.all(up[:69, :69] == 77)        # the cubic weights sum to one (up to float32 rounding, gone after rint)


def test_render_lists_follow_visualize():
    """predict.py:265-299: one disc per peak in part order; one ellipse per limb (not 9, 13) and
    person that has both joints, with ellipse2Poly's integer arguments."""
    from poseestimation_amd.predict_decode import render_lists
    peaks = [[] for _ in range(18)]
    peaks[1] = [(10, 20, 0.9, 0)]        # mode D tuples: (x, y, score, id)
    peaks[8] = [(40, 60, 0.8, 1)]
    peaks[2] = [(13, 20, 0.8, 2)]
    peaks[16] = [(17, 5, 0.8, 3)]
    cand = np.array([[10, 20, 0.9, 0], [40, 60, 0.8, 1], [13, 20, 0.8, 2], [17, 5, 0.8, 3]], float)
    person = -np.ones(20)
    person[[1, 8, 2, 16]] = [0, 1, 2, 3]
    circles, ellipses = render_lists(cand, peaks, person[None])
    assert circles.tolist() == [[10, 20, 1], [13, 20, 2], [40, 60, 8], [17, 5, 16]]
    # limb 0 = parts 1 -> 8: centre (25, 40), half length int(50 / 2), angle of (dy, dx) = (-40, -30) from b to a
    # limb 6 = parts 1 -> 2: centre (int(11.5), 20), half length int(1.5) = 1, angle atan2(0, -3) = 180
    # limb 9 = parts 2 -> 16 is skipped
    import math
    assert ellipses.tolist() == [[25, 40, 25, 4, int(math.degrees(math.atan2(-40, -30))), 0],
                                 [11, 20, 1, 4, 180, 6]]


def test_bench_pairs_a_context_with_few_buffers_whatever_the_run_lengths():
    """bench.py picks the context and the input buffer of a decode by one count of decodes
    issued since the start: whatever the lengths of the runs before it (priming, a warm-up of 5
    steps, a timed region of 20), a context meets at most six distinct buffers, so that it
    replays the launch sequences it has recorded (a context keeps eight)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        'bench_module', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for n_ctx, nbuf0 in ((8, 12), (4, 12), (4, 4), (1, 12), (6, 12)):
        nbuf = bench.graph_friendly_buffers(nbuf0, n_ctx)
        assert 1 <= nbuf <= nbuf0
        seen = {}
        count = 0
        for run_length in (3 * max(nbuf0, 2 * n_ctx), 5, 20, 7, 200, 1):
            for _ in range(run_length):
                seen.setdefault(count % n_ctx, set()).add(count % nbuf)
                count += 1
        assert max(len(v) for v in seen.values()) <= 6, (n_ctx, nbuf, seen)


@pytest.mark.parametrize('l,crop,L', [(23, 184, 368), (46, 368, 368), (69, 552, 368), (92, 736, 368),
                                      (26, 203, 97), (13, 97, 97), (58, 460, 200), (40, 313, 361)])
def test_multiscale_composite_operator_is_the_reference_pipeline(l, crop, L):
    """The sparse multi-scale peak stage evaluates Gaussian o resize o x8-resize as one banded
    operator per axis and scale.  The matrix the library builds (pd_test_ms_axis, host only) must
    be that pipeline: applied to random 1-D signals it reproduces oracle resize_cubic twice +
    the restated scipy filter; its cell form holds exactly its float32 weights, windows of at
    most 16 sources, and bound weights that dominate every own output of a cell."""
    import ctypes as C
    from poseestimation_amd import _ffi
    from oracle import decode_ref as ref
    Lb = _ffi.lib()
    identity = int(crop == L)
    nc = (L + 7) // 8
    K = np.zeros((L, l), np.float64)
    cw = np.zeros((nc, 10, 16), np.float32)
    bw = np.zeros((nc, 16), np.float32)
    lo = np.zeros((nc,), np.int32)
    factor = 0.25
    widest = Lb.pd_test_ms_axis(l, crop, L, identity, factor, K.ctypes.data, cw.ctypes.data, bw.ctypes.data, lo.ctypes.data)
    assert 1 <= widest <= 16
    rng = np.random.RandomState(l * 1000 + L)
    w = ref.gaussian_weights(3.0)
    for trial in range(3):
        src = rng.randn(l).astype(np.float32)
        up = ref.resize_cubic(src.reshape(l, 1, 1), fx=1, fy=8)[:crop]              # predict.py:63-64 along one axis
        if not identity:
            up = ref.resize_cubic(np.ascontiguousarray(up), dst_hw=(L, 1))          # predict.py:65
        want = ref._correlate_axis0(up[:, 0, 0].astype(np.float64), w) * factor
        got = K @ src.astype(np.float64)
        # (the reference rounds to float32 after every pass, the operator is its exact-arithmetic form)
        assert np.max(np.abs(got - want)) <= 4e-6 * max(1.0, np.abs(src).max())
    # the cell form: the same weights, rounded to float32, inside windows of 16
    dense = np.zeros((L, l), np.float32)
    for c in range(nc):
        for j in range(10):
            o = 8 * c - 1 + j
            if 0 <= o < L:
                for p in range(16):
                    if cw[c, j, p] != 0:
                        assert lo[c] + p < l
                        dense[o, lo[c] + p] = cw[c, j, p]
    for o in range(L):
        c, j = o // 8, o % 8 + 1
        row = np.zeros(l, np.float32)
        row[lo[c]:lo[c] + 16] = cw[c, j, :min(16, l - lo[c])]
        assert np.array_equal(row, K[o].astype(np.float32)), o
    assert np.array_equal(bw, np.abs(cw[:, 1:9, :]).max(axis=1))
    # a scale beyond what the sparse stage takes is refused, not truncated
    assert Lb.pd_test_ms_axis(200, 1600, 368, 0, 1.0, None, None, None, None) == -1
