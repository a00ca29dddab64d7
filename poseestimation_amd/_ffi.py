"""ctypes binding of libposedecode.so (the C ABI declared in include/posedecode.h).

There is no CPU fallback: if the library is missing or no GPU is visible the
decode raises.  ctypes releases the GIL for the duration of every call.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PD_LIBRARY') or os.path.join(_HERE, 'libposedecode.so')   # PD_LIBRARY: another build of it

PD_OK, PD_E_INVALID, PD_E_HIP, PD_E_OVERFLOW, PD_E_INDEX, PD_E_STATE = 0, -1, -2, -3, -4, -5
MODE_P, MODE_D = 0, 1
MEM_HOST, MEM_DEVICE = 0, 1
F32, F64, U8 = 0, 1, 2
IMPL_FAST, IMPL_SIMPLE, IMPL_STREAM = 0, 1, 2

EXPORTS = ['pd_create', 'pd_destroy', 'pd_last_error', 'pd_version', 'pd_set_mode', 'pd_set_impl',
           'pd_stream', 'pd_share_stream', 'pd_upsample_avg', 'pd_set_multiscale', 'pd_decode_multiscale', 'pd_read_fullres', 'pd_load_fullres', 'pd_set_lowres',
           'pd_peaks', 'pd_set_peaks', 'pd_connections', 'pd_set_connections', 'pd_parse',
           'pd_decode_batch', 'pd_decode_fullres', 'pd_pack', 'pd_sync_totals', 'pd_fetch',
           'pd_debug_smoothed', 'pd_packed_device', 'pd_profile', 'pd_profile_read', 'pd_results_host', 'pd_filter_stats', 'pd_profile_stages', 'pd_set_skeleton_buffer', 'pd_submit', 'pd_flush', 'pd_set_graph_replay', 'pd_set_host_results',
           'pd_decode_batch_static', 'pd_coco_rows', 'pd_image_resize_shape', 'pd_image_resize_pad', 'pd_render',
           'pd_debug_found', 'pd_test_ms_axis', 'pd_comm_unique_id', 'pd_comm_create', 'pd_comm_destroy', 'pd_comm_last_error', 'pd_gather', 'pd_copy_used_skeletons']

STAGES = ['resize', 'smooth_nms', 'sort', 'pairs', 'match', 'parse', 'pack', 'filter', 'verify', 'planar', 'cells']


class Config(C.Structure):
    _fields_ = [('device_id', C.c_int32), ('max_batch', C.c_int32), ('max_h', C.c_int32),
                ('max_w', C.c_int32), ('max_peaks_per_part', C.c_int32), ('max_persons', C.c_int32),
                ('max_pairs_per_image', C.c_int32), ('fullres', C.c_int32)]


class Totals(C.Structure):
    _fields_ = [('n_images', C.c_int32), ('first_bad_image', C.c_int32), ('n_peaks', C.c_int64),
                ('n_conns', C.c_int64), ('n_persons', C.c_int64)]


PEAK_DTYPE = np.dtype([('y', np.int32), ('x', np.int32), ('score', np.float64)])
CONN_DTYPE = np.dtype([('id_a', np.int32), ('id_b', np.int32), ('ia', np.int32), ('ib', np.int32),
                       ('score', np.float64)])


def res_layout(B):
    """Byte offsets inside the packed result buffer (ResLayout in csrc/common.h)."""
    off_part = 64
    off_limb = off_part + B * 18 * 4
    off_pers = off_limb + B * 19 * 4
    off_stat = off_pers + B * 4
    off_payload = (off_stat + B * 4 + 15) & ~15
    return off_part, off_limb, off_pers, off_stat, off_payload


def unpack_results(raw, B=None):
    """uint8 image of the packed result buffer -> dict of numpy views."""
    head = raw[:64].view(np.int64)
    if B is None:
        B = int(head[0])
    npk, ncn, nps = int(head[1]), int(head[2]), int(head[3])
    off_part, off_limb, off_pers, off_stat, off_pay = res_layout(B)
    p1 = off_pay + npk * PEAK_DTYPE.itemsize
    p2 = p1 + ncn * CONN_DTYPE.itemsize
    p3 = p2 + nps * 160
    return {
        'B': B,
        'part_count': raw[off_part:off_limb].view(np.int32).reshape(B, 18),
        'limb_count': raw[off_limb:off_pers].view(np.int32).reshape(B, 19),
        'person_count': raw[off_pers:off_stat].view(np.int32),
        'status': raw[off_stat:off_stat + 4 * B].view(np.int32),
        'peaks': raw[off_pay:p1].view(PEAK_DTYPE),
        'conns': raw[p1:p2].view(CONN_DTYPE),
        'persons': raw[p2:p3].view(np.float64).reshape(nps, 20),
        'raw': raw[:p3],
    }


class DecodeOverflow(RuntimeError):
    """A capacity given at context creation was exceeded (PD_E_OVERFLOW)."""


_lib = None


def lib():
    """Load the shared library once; raise if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError('libposedecode.so is not built (%s); run `python -c "import __graft_entry__ as g; '
                           'g.build()"` or `make -C poseestimation_amd/csrc`' % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, i32, dbl = C.c_void_p, C.c_int, C.c_double
    L.pd_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.pd_destroy.argtypes = [vp]
    L.pd_last_error.argtypes = [vp]
    L.pd_last_error.restype = C.c_char_p
    L.pd_version.restype = C.c_char_p
    L.pd_set_mode.argtypes = [vp, i32]
    L.pd_set_impl.argtypes = [vp, i32]
    L.pd_stream.argtypes = [vp]
    L.pd_stream.restype = vp
    L.pd_share_stream.argtypes = [vp, vp]
    L.pd_upsample_avg.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                  i32, i32, i32, i32]
    L.pd_set_multiscale.argtypes = L.pd_upsample_avg.argtypes
    L.pd_decode_multiscale.argtypes = L.pd_upsample_avg.argtypes[:-1] + [dbl, dbl, i32]
    L.pd_read_fullres.argtypes = [vp, vp, vp]
    L.pd_load_fullres.argtypes = [vp, vp, i32, i32, vp, i32, i32, i32, i32, i32, i32]
    L.pd_set_lowres.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, i32]
    L.pd_peaks.argtypes = [vp, dbl]
    L.pd_set_peaks.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pd_connections.argtypes = [vp, dbl]
    L.pd_set_connections.argtypes = [vp, vp, vp]
    L.pd_parse.argtypes = [vp]
    L.pd_decode_batch.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, dbl, dbl, i32]
    L.pd_decode_fullres.argtypes = [vp, dbl, dbl]
    L.pd_pack.argtypes = [vp]
    L.pd_sync_totals.argtypes = [vp, C.POINTER(Totals)]
    L.pd_fetch.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.pd_debug_smoothed.argtypes = [vp, vp]
    L.pd_packed_device.argtypes = [vp] + [C.POINTER(vp)] * 6
    L.pd_results_host.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.pd_profile.argtypes = [vp, i32]
    L.pd_profile_stages.argtypes = [vp, C.c_uint32]
    L.pd_filter_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.pd_set_skeleton_buffer.argtypes = [vp, vp, C.c_uint64]
    L.pd_submit.argtypes = [vp, vp, vp, i32, i32, i32, i32, i32, dbl, dbl, vp, C.c_uint64]
    L.pd_flush.argtypes = [vp]
    L.pd_set_graph_replay.argtypes = [vp, i32]
    L.pd_set_host_results.argtypes = [vp, i32]
    L.pd_profile_read.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    L.pd_decode_batch_static.argtypes = [vp, vp, i32, i32, i32, i32, i32, dbl, dbl, i32, vp, C.c_uint64, vp]
    i32p = C.POINTER(C.c_int32)
    L.pd_coco_rows.argtypes = [vp, vp, vp, vp, C.c_int64, C.POINTER(C.c_int64)]
    L.pd_image_resize_shape.argtypes = [i32, i32, dbl, i32, i32p, i32p, i32p, i32p]
    L.pd_image_resize_pad.argtypes = [vp, vp, i32, i32, i32, i32, dbl, i32, dbl, vp, i32, i32, i32, i32]
    L.pd_render.argtypes = [vp, vp, i32, i32, i32, vp, i32, vp, vp, vp]
    L.pd_debug_found.argtypes = [vp, i32, vp]
    L.pd_test_ms_axis.argtypes = [i32, i32, i32, i32, dbl, vp, vp, vp, vp]
    L.pd_comm_unique_id.argtypes = [vp]
    L.pd_comm_create.argtypes = [vp, i32, i32, i32, C.POINTER(vp)]
    L.pd_comm_destroy.argtypes = [vp]
    L.pd_comm_last_error.argtypes = [vp]
    L.pd_comm_last_error.restype = C.c_char_p
    L.pd_gather.argtypes = [vp, C.POINTER(vp), i32, i32, vp, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.pd_copy_used_skeletons.argtypes = [vp, vp, i32, C.c_uint64, vp]
    for name in EXPORTS:
        fn = getattr(L, name)
        if name not in ('pd_last_error', 'pd_version', 'pd_stream', 'pd_comm_last_error'):
            fn.restype = C.c_int
    _lib = L
    return L


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Context:
    """Owns one pd_ctx (one GPU, one HIP stream)."""

    def __init__(self, device_id=0, max_batch=1, max_h=368, max_w=368, max_peaks_per_part=128,
                 max_persons=128, max_pairs_per_image=32768, fullres=False):
        self._h = None
        L = lib()
        cfg = Config(device_id, max_batch, max_h, max_w, max_peaks_per_part, max_persons,
                     max_pairs_per_image, 1 if fullres else 0)
        h = C.c_void_p()
        rc = L.pd_create(C.byref(cfg), C.byref(h))
        if rc != PD_OK:
            raise RuntimeError('pd_create failed (%d): %s' % (rc, L.pd_last_error(None).decode()))
        self._h = h
        self.cfg = cfg
        self.L = L

    def close(self):
        if self._h is not None:
            self.L.pd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- error mapping: ValueError for bad shapes, IndexError for the mode-D case --
    def check(self, rc):
        if rc == PD_OK:
            return
        msg = self.L.pd_last_error(self._h).decode()
        if rc == PD_E_INVALID:
            raise ValueError(msg)
        if rc == PD_E_INDEX:
            raise IndexError(msg)
        if rc == PD_E_OVERFLOW:
            raise DecodeOverflow(msg)
        raise RuntimeError('libposedecode error %d: %s' % (rc, msg))

    @property
    def stream(self):
        return self.L.pd_stream(self._h)

    def share_stream(self, owner):
        """Launch on ``owner``'s stream from now on (see pd_share_stream); keeps ``owner`` alive."""
        self.check(self.L.pd_share_stream(self._h, owner._h))
        self._stream_owner = owner

    def set_mode(self, mode):
        self.check(self.L.pd_set_mode(self._h, mode))

    def set_impl(self, impl):
        self.check(self.L.pd_set_impl(self._h, impl))

    def set_lowres(self, heat, paf, B, h, w, H, W, mem=MEM_HOST):
        self.check(self.L.pd_set_lowres(self._h, heat, paf, B, h, w, H, W, mem))

    def load_fullres(self, heat, paf, B, H, W):
        """heat / paf: C-contiguous (B, H, W, C) float32 or float64 numpy arrays or None."""
        def desc(a):
            if a is None:
                return None, F64, 0
            return C.c_void_p(a.ctypes.data), (F32 if a.dtype == np.float32 else F64), a.shape[-1]
        hp, hd, hc = desc(heat)
        pp, pdt, pc = desc(paf)
        self.check(self.L.pd_load_fullres(self._h, hp, hd, hc, pp, pdt, pc, B, H, W, MEM_HOST))

    def upsample_avg(self, heat_list, paf_list, crops, H, W, B):
        n = len(heat_list)
        hp = (C.c_void_p * n)(*[a.ctypes.data for a in heat_list])
        pp = (C.c_void_p * n)(*[a.ctypes.data for a in paf_list])
        hs = (C.c_int32 * n)(*[a.shape[1] for a in heat_list])
        ws = (C.c_int32 * n)(*[a.shape[2] for a in heat_list])
        ch = (C.c_int32 * n)(*[c[0] for c in crops])
        cw = (C.c_int32 * n)(*[c[1] for c in crops])
        self.check(self.L.pd_upsample_avg(self._h, n, hp, pp, hs, ws, ch, cw, H, W, B, MEM_HOST))

    def upsample_avg_device(self, heat_ptrs, paf_ptrs, sizes, crops, H, W, B):
        """The same for device-resident network outputs: addresses of the (B, hs, ws, 19) and
        (B, hs, ws, 38) float32 arrays of every scale, ``sizes[m] = (hs, ws)``."""
        n = len(heat_ptrs)
        hp = (C.c_void_p * n)(*heat_ptrs)
        pp = (C.c_void_p * n)(*paf_ptrs)
        hs = (C.c_int32 * n)(*[s[0] for s in sizes])
        ws = (C.c_int32 * n)(*[s[1] for s in sizes])
        ch = (C.c_int32 * n)(*[c[0] for c in crops])
        cw = (C.c_int32 * n)(*[c[1] for c in crops])
        self.check(self.L.pd_upsample_avg(self._h, n, hp, pp, hs, ws, ch, cw, H, W, B, MEM_DEVICE))

    @staticmethod
    def _scale_args(heat_ptrs, paf_ptrs, sizes, crops):
        n = len(heat_ptrs)
        return (n, (C.c_void_p * n)(*heat_ptrs), (C.c_void_p * n)(*paf_ptrs),
                (C.c_int32 * n)(*[s[0] for s in sizes]), (C.c_int32 * n)(*[s[1] for s in sizes]),
                (C.c_int32 * n)(*[c[0] for c in crops]), (C.c_int32 * n)(*[c[1] for c in crops]))

    def set_multiscale(self, heat_ptrs, paf_ptrs, sizes, crops, H, W, B, mem=MEM_DEVICE):
        """The network outputs of every test scale as the source of the stages that follow
        (nothing at full resolution is stored): addresses of the (B, hs, ws, 19) / (B, hs, ws, 38)
        float32 arrays, ``sizes[m] = (hs, ws)``, ``crops[m]`` = the part of the x8 map kept."""
        self.check(self.L.pd_set_multiscale(self._h, *self._scale_args(heat_ptrs, paf_ptrs, sizes, crops), H, W, B, mem))

    def decode_multiscale(self, heat_ptrs, paf_ptrs, sizes, crops, H, W, B, thre1, thre2, mem=MEM_DEVICE):
        """predict.py:49-254 behind the network calls in one enqueue (see pd_decode_multiscale)."""
        self.check(self.L.pd_decode_multiscale(self._h, *self._scale_args(heat_ptrs, paf_ptrs, sizes, crops), H, W, B,
                                               float(thre1), float(thre2), mem))

    def read_fullres(self, B, H, W, want_heat=True, want_paf=True):
        heat = np.empty((B, H, W, 19), np.float64) if want_heat else None
        paf = np.empty((B, H, W, 38), np.float64) if want_paf else None
        self.check(self.L.pd_read_fullres(self._h, _ptr(heat), _ptr(paf)))
        return heat, paf

    def peaks(self, thre1):
        self.check(self.L.pd_peaks(self._h, float(thre1)))

    def set_peaks(self, counts, y, x, score, ids):
        counts = np.ascontiguousarray(counts, np.int32)
        y = np.ascontiguousarray(y, np.int32)
        x = np.ascontiguousarray(x, np.int32)
        score = np.ascontiguousarray(score, np.float64)
        ids = np.ascontiguousarray(ids, np.int32)
        self.check(self.L.pd_set_peaks(self._h, _ptr(counts), _ptr(y), _ptr(x), _ptr(score), _ptr(ids)))

    def connections(self, thre2):
        self.check(self.L.pd_connections(self._h, float(thre2)))

    def set_connections(self, counts, rows):
        counts = np.ascontiguousarray(counts, np.int32)
        rows = np.ascontiguousarray(rows, np.float64)
        self.check(self.L.pd_set_connections(self._h, _ptr(counts), _ptr(rows)))

    def parse(self):
        self.check(self.L.pd_parse(self._h))

    def decode_batch(self, heat, paf, B, h, w, H, W, thre1, thre2, mem=MEM_HOST):
        self.check(self.L.pd_decode_batch(self._h, heat, paf, B, h, w, H, W, float(thre1), float(thre2), mem))

    def decode_fullres(self, thre1, thre2):
        self.check(self.L.pd_decode_fullres(self._h, float(thre1), float(thre2)))

    def pack(self):
        self.check(self.L.pd_pack(self._h))

    def sync_totals(self):
        t = Totals()
        self.check(self.L.pd_sync_totals(self._h, C.byref(t)))
        return t

    def fetch(self):
        """Packed results of the stages that ran, as numpy arrays (one copy out of
        the context's pinned result buffer)."""
        t = self.sync_totals()      # waits; raises on overflow / mode-D index error
        base = C.c_void_p()
        nbytes = C.c_uint64()
        self.check(self.L.pd_results_host(self._h, C.byref(base), C.byref(nbytes)))
        raw = np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(nbytes.value,)).copy()
        return unpack_results(raw, t.n_images)

    def fetch_via_pd_fetch(self):
        """The same through pd_fetch (caller-allocated arrays)."""
        t = self.sync_totals()
        B = t.n_images
        out = {
            'B': B,
            'part_count': np.zeros((B, 18), np.int32),
            'limb_count': np.zeros((B, 19), np.int32),
            'person_count': np.zeros((B,), np.int32),
            'status': np.zeros((B,), np.int32),
            'peaks': np.zeros((t.n_peaks,), PEAK_DTYPE),
            'conns': np.zeros((t.n_conns,), CONN_DTYPE),
            'persons': np.zeros((t.n_persons, 20), np.float64),
        }
        self.check(self.L.pd_fetch(self._h, _ptr(out['part_count']), _ptr(out['limb_count']),
                                   _ptr(out['person_count']), _ptr(out['status']), _ptr(out['peaks']),
                                   _ptr(out['conns']), _ptr(out['persons'])))
        return out

    def profile(self, enable=True, stages=None):
        """Time every stage (enable=True), none, or only the named stages."""
        if stages is not None:
            mask = 0
            for name in stages:
                mask |= 1 << STAGES.index(name)
            self.check(self.L.pd_profile_stages(self._h, mask))
        else:
            self.check(self.L.pd_profile(self._h, 1 if enable else 0))

    def profile_read(self):
        """{stage: (total_ms, launches)} since the last read (waits for the stream)."""
        ms = (C.c_double * len(STAGES))()
        n = (C.c_int64 * len(STAGES))()
        self.check(self.L.pd_profile_read(self._h, ms, n))
        return {s: (ms[i], n[i]) for i, s in enumerate(STAGES)}

    def packed_device(self):
        """Device addresses of the packed results: peaks, conns, persons, part_count,
        limb_count, person_count."""
        ptrs = [C.c_void_p() for _ in range(6)]
        self.check(self.L.pd_packed_device(self._h, *[C.byref(p) for p in ptrs]))
        return [p.value for p in ptrs]

    def set_graph_replay(self, enable):
        self.check(self.L.pd_set_graph_replay(self._h, 1 if enable else 0))

    def set_skeleton_buffer(self, device_ptr, capacity):
        """Device buffer that receives peaks + person rows (no connections) with every pack."""
        self.check(self.L.pd_set_skeleton_buffer(self._h, device_ptr, capacity))

    def set_host_results(self, enable):
        """False: with a skeleton buffer set, decodes write that buffer only (nothing to fetch)."""
        self.check(self.L.pd_set_host_results(self._h, 1 if enable else 0))

    def submit(self, heat_ptr, paf_ptr, B, h, w, H, W, thre1, thre2, skeleton_ptr=None, skeleton_capacity=0):
        """Queue a decode of device-resident inputs for the context's launcher thread."""
        self.check(self.L.pd_submit(self._h, heat_ptr, paf_ptr, B, h, w, H, W, thre1, thre2, skeleton_ptr,
                                    skeleton_capacity))

    def flush(self):
        """Wait until every queued decode has been launched; raises its first error."""
        self.check(self.L.pd_flush(self._h))

    def sync(self):
        """Wait for everything queued on the context's stream.  Raises on a decode error when
        the context packs host results; a skeleton-only context (``set_host_results(False)``)
        reports failed images through the skeleton buffer (``SkeletonGather.results`` raises)."""
        self.check(self.L.pd_sync_totals(self._h, None))

    def filter_stats(self):
        """Counters of the last filtered peak stage: candidates, interior cells, border
        cells, planes redone by the exact kernel."""
        out = (C.c_int64 * 4)()
        self.check(self.L.pd_filter_stats(self._h, out))
        return dict(candidates=out[0], cells_interior=out[1], cells_border=out[2], planes_exact=out[3])

    def debug_found(self, enable=True, read=False):
        """Recording of ``len(found)`` per limb-18 connection (utils/postprocess.py:224-225);
        ``read=True`` returns the (max_batch, max_peaks_per_part) int32 table first."""
        out = None
        if read:
            out = np.zeros((self.cfg.max_batch, self.cfg.max_peaks_per_part), np.int32)
        self.check(self.L.pd_debug_found(self._h, 1 if enable else 0, _ptr(out)))
        return out

    def coco_rows(self):
        """evaluate.py:44-74 on the device: ``(keypoints (n, 17, 3), scores (n,), image_index (n,))``
        for the person rows of the decode the context holds."""
        cap = max(1, int(self.cfg.max_batch) * int(self.cfg.max_persons))
        n = C.c_int64()
        # the row count is only known on the device: ask first with no room, then fetch
        rc = self.L.pd_coco_rows(self._h, None, None, None, 0, C.byref(n))
        if rc not in (PD_OK, PD_E_OVERFLOW):
            self.check(rc)
        cap = int(n.value)
        kp = np.zeros((cap, 17, 3), np.float64)
        sc = np.zeros((cap,), np.float64)
        im = np.zeros((cap,), np.int32)
        if cap:
            self.check(self.L.pd_coco_rows(self._h, _ptr(kp), _ptr(sc), _ptr(im), cap, C.byref(n)))
        return kp, sc, im

    def image_resize_shape(self, h, w, scale, stride):
        """``(resized_h, resized_w, pad_down, pad_right)`` of predict.py:56-57."""
        v = [C.c_int32() for _ in range(4)]
        rc = self.L.pd_image_resize_shape(h, w, float(scale), stride, *[C.byref(x) for x in v])
        if rc != PD_OK:
            raise ValueError('image_resize_shape: bad size %dx%d, scale %r or stride %d' % (h, w, scale, stride))
        return tuple(x.value for x in v)

    def image_resize_pad(self, img, scale, stride, pad_value, out_dtype=None):
        """predict.py:56-57 on the device: cubic resize by ``scale``, then padding below / right
        to a multiple of ``stride``.  ``img`` (h, w, C) uint8 or float32; returns
        ``(padded (oh, ow, C), [0, 0, pad_down, pad_right])``."""
        img = np.ascontiguousarray(img)
        if img.ndim != 3 or img.dtype not in (np.uint8, np.float32):
            raise ValueError('img must be (h, w, C) uint8 or float32, got %r %s' % (img.shape, img.dtype))
        out_dtype = np.dtype(out_dtype or img.dtype)
        h, w, ch = img.shape
        rh, rw, pd, pr = self.image_resize_shape(h, w, scale, stride)
        out = np.empty((rh + pd, rw + pr, ch), out_dtype)
        code = {np.dtype(np.uint8): U8, np.dtype(np.float32): F32}
        self.check(self.L.pd_image_resize_pad(self._h, _ptr(img), code[img.dtype], h, w, ch, float(scale), stride,
                                              float(pad_value), _ptr(out), code[out_dtype], rh + pd, rw + pr, MEM_HOST))
        return out, [0, 0, pd, pr]

    def render(self, img, circles, ellipses, want_part=True, want_final=True):
        """predict.py:256-307: ``circles`` (n, 3) int32 ``(x, y, colour)``, ``ellipses`` (m, 6) int32
        ``(cx, cy, a, b, angle_degrees, colour)``; returns ``(part_image, final_image)`` uint8."""
        img = np.ascontiguousarray(img, np.uint8)
        if img.ndim != 3 or img.shape[2] != 3:
            raise ValueError('img must be (H, W, 3) uint8')
        circles = np.ascontiguousarray(circles, np.int32).reshape(-1, 3)
        ellipses = np.ascontiguousarray(ellipses, np.int32).reshape(-1, 6)
        part = np.empty_like(img) if want_part else None
        final = np.empty_like(img) if want_final else None
        self.check(self.L.pd_render(self._h, _ptr(img), img.shape[0], img.shape[1], len(circles),
                                    _ptr(circles) if len(circles) else None, len(ellipses),
                                    _ptr(ellipses) if len(ellipses) else None, _ptr(part), _ptr(final)))
        return part, final

    def debug_smoothed(self, B, H, W):
        out = np.empty((B, 18, H, W), np.float64)
        self.check(self.L.pd_debug_smoothed(self._h, _ptr(out)))
        return out


class Comm:
    """A communicator of the library's own (RCCL loaded at run time, no framework needed):
    ``Comm.unique_id()`` on one rank, the 128 bytes handed to the others, then
    ``Comm(id, world, rank, device_id)`` on every rank."""

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        L = lib()
        rc = L.pd_comm_unique_id(buf)
        if rc != PD_OK:
            raise RuntimeError('pd_comm_unique_id failed (%d): %s' % (rc, L.pd_last_error(None).decode()))
        return bytes(buf)

    def __init__(self, unique_id, world, rank, device_id=0):
        self._h = None
        self.L = lib()
        self.world, self.rank = world, rank
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        rc = self.L.pd_comm_create(buf, world, rank, device_id, C.byref(h))
        if rc != PD_OK:
            raise RuntimeError('pd_comm_create failed (%d): %s' % (rc, self.L.pd_last_error(None).decode()))
        self._h = h

    def close(self):
        if self._h is not None:
            self.L.pd_comm_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def gather(self, contexts, root=0, capacity=1 << 22):
        """pd_gather: the skeleton buffers of every rank's contexts on ``root``.  Returns, on the
        root, a list (rank-major, then context order) of uint8 arrays in the packed result
        layout (``unpack_results`` reads them); None elsewhere."""
        n = len(contexts)
        arr = (C.c_void_p * n)(*[c._h for c in contexts])
        is_root = self.rank == root
        recv = np.zeros(capacity if is_root else 1, np.uint8)
        sizes = (C.c_uint64 * (self.world * n))()
        total = C.c_uint64()
        rc = self.L.pd_gather(self._h, arr, n, root, _ptr(recv), capacity if is_root else 0, sizes, C.byref(total))
        if rc != PD_OK:
            msg = self.L.pd_comm_last_error(self._h).decode()
            if rc == PD_E_OVERFLOW:
                raise DecodeOverflow(msg)
            raise RuntimeError('pd_gather failed (%d): %s' % (rc, msg))
        if not is_root:
            return None
        out, off = [], 0
        for v in sizes:
            out.append(recv[off:off + int(v)])
            off += (int(v) + 15) & ~15
        return out
