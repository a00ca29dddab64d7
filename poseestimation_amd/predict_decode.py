"""The decode half of the reference's ``predict.detect`` (predict.py:45-254) on the GPU.

* ``pad_right_down_corner``  predict.py:19-32
* ``upsample_average``       predict.py:51-72   (x8 bicubic, un-pad, resize, multi-scale average)
* ``decode``                 predict.py:82-254  -> ``(candidate, all_peaks, subset)``
* ``decode_scales``          predict.py:63-254 from the network outputs of the scales, no maps stored
* ``DecodeEngine``           the batched form: many images per call, results packed
* ``resize_pad``             predict.py:56-57  (image-side cubic resize + padding, on the device)
* ``detect``                 predict.py:45 signature; the network is the caller's ``predict_func``
* ``visualize``              predict.py:256-307 (peak discs and limb ellipses, rasterised on the device)

``mode='D'`` follows predict.py (peak tuples ``(x, y, score, id)``, PAF channel 2k with
dx, sample index rounded half-even, connection rows ``[id_a, id_b, score, i, j]``);
``mode='P'`` follows utils/postprocess.py.
"""
import numpy as np

from . import _ffi
from . import postprocess as _pp
from .config import (IMG_Y, PAD_VALUE, SCALE_SEARCH, STRIDE, THRE1, THRE2)

_MODES = {'P': _ffi.MODE_P, 'D': _ffi.MODE_D, _ffi.MODE_P: _ffi.MODE_P, _ffi.MODE_D: _ffi.MODE_D}


def pad_right_down_corner(img, stride=STRIDE, pad_value=PAD_VALUE):
    """predict.py:19-32: pad below / right to a multiple of ``stride``;
    returns ``(img_padded, [0, 0, down, right])``."""
    h, w = img.shape[:2]
    pad = [0, 0, (stride - h % stride) % stride, (stride - w % stride) % stride]
    out = np.full((h + pad[2], w + pad[3]) + img.shape[2:], pad_value, dtype=img.dtype)
    out[:h, :w] = img
    return out, pad


class Results:
    """Packed decode results of a batch with per-image accessors that rebuild the
    reference's Python structures."""

    def __init__(self, packed, mode):
        self.packed = _pp._with_starts(packed)
        self.mode = mode
        self.B = packed['B']

    def all_peaks(self, b, score_dtype=np.float64):
        return _pp._peaks_from_packed(self.packed, b, score_dtype, xy_order=self.mode == _ffi.MODE_D)

    def candidate(self, b):
        """predict.py:179: the flattened peak list as an (n, 4) float64 array."""
        s, e = int(self.packed['peak_start'][b]), int(self.packed['peak_start'][b + 1])
        seg = self.packed['peaks'][s:e]
        out = np.empty((e - s, 4))
        if self.mode == _ffi.MODE_D:
            out[:, 0], out[:, 1] = seg['x'], seg['y']
        else:
            out[:, 0], out[:, 1] = seg['y'], seg['x']
        out[:, 2] = seg['score']
        out[:, 3] = np.arange(e - s)
        return out

    def connections(self, b):
        return _pp._conns_from_packed(self.packed, b, self.all_peaks(b), self.mode)

    def subset(self, b):
        s, e = int(self.packed['person_start'][b]), int(self.packed['person_start'][b + 1])
        return self.packed['persons'][s:e].copy()

    def image(self, b):
        """``(candidate, all_peaks, subset)`` as predict.detect returns them."""
        return self.candidate(b), self.all_peaks(b), self.subset(b)


class DecodeEngine:
    """Batched decode on one GPU.  ``decode_batch`` takes network outputs
    ``heat (B, h, w, 19)`` / ``paf (B, h, w, 38)`` float32 (host numpy arrays, or
    device pointers with ``device=True``) and returns :class:`Results`."""

    def __init__(self, max_batch, max_h=368, max_w=368, mode='D', device_id=0, max_peaks_per_part=128,
                 max_persons=128, max_pairs_per_image=32768, fullres=False, share_stream_with=None):
        """``share_stream_with``: another engine of the same GPU whose HIP stream this one uses
        too; the two, used in turn by one thread, keep that stream two decodes deep."""
        self.mode = _MODES[mode]
        self.ctx = _ffi.Context(device_id=device_id, max_batch=max_batch, max_h=max_h, max_w=max_w,
                                max_peaks_per_part=max_peaks_per_part, max_persons=max_persons,
                                max_pairs_per_image=max_pairs_per_image, fullres=fullres)
        self.ctx.set_mode(self.mode)
        if share_stream_with is not None:
            self.ctx.share_stream(share_stream_with.ctx)

    def close(self):
        self.ctx.close()

    def enqueue(self, heat, paf, out_hw=None, thre1=THRE1, thre2=THRE2):
        """Launch the decode of a batch without waiting (numpy inputs)."""
        heat = np.ascontiguousarray(heat, np.float32)
        paf = np.ascontiguousarray(paf, np.float32)
        if heat.ndim != 4 or paf.ndim != 4 or heat.shape[3] != 19 or paf.shape[3] != 38 \
                or heat.shape[:3] != paf.shape[:3]:
            raise ValueError('heat must be (B, h, w, 19) and paf (B, h, w, 38), got %r / %r'
                             % (heat.shape, paf.shape))
        B, h, w = heat.shape[:3]
        H, W = out_hw if out_hw is not None else (h * STRIDE, w * STRIDE)
        self.ctx.decode_batch(heat.ctypes.data, paf.ctypes.data, B, h, w, H, W, thre1, thre2, _ffi.MEM_HOST)

    def enqueue_device(self, heat_ptr, paf_ptr, B, h, w, out_hw=None, thre1=THRE1, thre2=THRE2):
        """The same for device-resident float32 arrays given by address."""
        H, W = out_hw if out_hw is not None else (h * STRIDE, w * STRIDE)
        self.ctx.decode_batch(heat_ptr, paf_ptr, B, h, w, H, W, thre1, thre2, _ffi.MEM_DEVICE)

    def fetch(self):
        return Results(self.ctx.fetch(), self.mode)

    def decode_batch(self, heat, paf, out_hw=None, thre1=THRE1, thre2=THRE2):
        self.enqueue(heat, paf, out_hw, thre1, thre2)
        return self.fetch()

    def upsample_average(self, heat_scales, paf_scales, pads, out_hw):
        """predict.py:51-72 for a batch: ``heat_scales[m]`` is (B, hs, ws, 19)."""
        hs_list = [np.ascontiguousarray(a, np.float32) for a in heat_scales]
        ps_list = [np.ascontiguousarray(a, np.float32) for a in paf_scales]
        B = hs_list[0].shape[0]
        crops = [(a.shape[1] * STRIDE - p[0], a.shape[2] * STRIDE - p[1]) for a, p in zip(hs_list, pads)]
        H, W = out_hw
        self.ctx.upsample_avg(hs_list, ps_list, crops, H, W, B)
        return B, H, W

    def decode_multiscale(self, heat_scales, paf_scales, pads, out_hw, thre1=THRE1, thre2=THRE2, materialise=False):
        """predict.py:51-254 for a batch: the scale loop's average and the decode, on the device.
        The averaged maps are not stored (``materialise=True``: the older form, which writes
        them out with ``upsample_average`` and decodes those; same results)."""
        if materialise:
            self.upsample_average(heat_scales, paf_scales, pads, out_hw)
            self.ctx.decode_fullres(thre1, thre2)
            return self.fetch()
        hs_list = [np.ascontiguousarray(a, np.float32) for a in heat_scales]
        ps_list = [np.ascontiguousarray(a, np.float32) for a in paf_scales]
        B = hs_list[0].shape[0]
        sizes = [a.shape[1:3] for a in hs_list]
        crops = [(a.shape[1] * STRIDE - p[0], a.shape[2] * STRIDE - p[1]) for a, p in zip(hs_list, pads)]
        H, W = out_hw
        self.ctx.decode_multiscale([a.ctypes.data for a in hs_list], [a.ctypes.data for a in ps_list], sizes, crops,
                                   H, W, B, thre1, thre2, _ffi.MEM_HOST)
        return self.fetch()

    def enqueue_multiscale_device(self, heat_ptrs, paf_ptrs, sizes, crops, out_hw, B, thre1=THRE1, thre2=THRE2):
        """The same for device-resident network outputs given by address, without waiting."""
        H, W = out_hw
        self.ctx.decode_multiscale(heat_ptrs, paf_ptrs, sizes, crops, H, W, B, thre1, thre2, _ffi.MEM_DEVICE)


def upsample_average(net_outputs_per_scale, pads, out_hw):
    """predict.py:51-72 for one image.  ``net_outputs_per_scale`` is a list of
    ``(heatmap, paf)`` pairs, each ``(1, hs, ws, C)`` or ``(hs, ws, C)`` float32, one
    per test scale; ``pads[m]`` is the ``pad`` list of pad_right_down_corner (or a
    ``(down, right)`` pair).  Returns ``(heatmap_avg (H, W, 19), paf_avg (H, W, 38))``
    float64."""
    heats, pafs = [], []
    for hm, pf in net_outputs_per_scale:
        hm = np.asarray(hm, np.float32)
        pf = np.asarray(pf, np.float32)
        heats.append(hm[None] if hm.ndim == 3 else hm)
        pafs.append(pf[None] if pf.ndim == 3 else pf)
    pads2 = [(p[2], p[3]) if len(p) == 4 else tuple(p) for p in pads]
    H, W = out_hw
    ctx = _pp._context(H, W, 128, 32768, 128)       # the cached context of this process
    crops = [(a.shape[1] * STRIDE - p[0], a.shape[2] * STRIDE - p[1]) for a, p in zip(heats, pads2)]
    ctx.upsample_avg(heats, pafs, crops, H, W, 1)
    heat, paf = ctx.read_fullres(1, H, W)
    return heat[0], paf[0]


def decode_scales(net_outputs_per_scale, pads, out_hw, thre1=THRE1, thre2=THRE2, mode='D'):
    """predict.py:63-254 for one image from the network outputs of the test scales:
    ``(candidate, all_peaks, subset)``, the result of ``decode(*upsample_average(..))`` without
    the averaged maps being stored, read back or uploaded again (the decode evaluates them on
    the device where it reads them)."""
    m = _MODES[mode]
    heats, pafs = [], []
    for hm, pf in net_outputs_per_scale:
        hm = np.ascontiguousarray(hm, np.float32)
        pf = np.ascontiguousarray(pf, np.float32)
        heats.append(hm[None] if hm.ndim == 3 else hm)
        pafs.append(pf[None] if pf.ndim == 3 else pf)
    pads2 = [(p[2], p[3]) if len(p) == 4 else tuple(p) for p in pads]
    H, W = out_hw
    sizes = [a.shape[1:3] for a in heats]
    crops = [(a.shape[1] * STRIDE - p[0], a.shape[2] * STRIDE - p[1]) for a, p in zip(heats, pads2)]
    pk, pairs = 128, 32768
    while True:
        ctx = _pp._context(H, W, pk, pairs, 128)
        ctx.set_mode(m)
        ctx.decode_multiscale([a.ctypes.data for a in heats], [a.ctypes.data for a in pafs], sizes, crops, H, W, 1,
                              thre1, thre2, _ffi.MEM_HOST)
        try:
            res = Results(ctx.fetch(), m)
            break
        except _ffi.DecodeOverflow:
            if pk >= 4096:
                raise
            pk, pairs = pk * 4, pairs * 16
    return res.image(0)


def decode(heatmap_avg, paf_avg, thre1=THRE1, thre2=THRE2, mode='D'):
    """predict.py:82-254: peaks, connections and person rows of one image from
    full-resolution maps.  Returns ``(candidate, all_peaks, subset)``."""
    m = _MODES[mode]
    hm = _pp._as_float_array(heatmap_avg, 18, 'heatmap_avg')
    pf = _pp._as_float_array(paf_avg, 38, 'paf_avg')
    if hm.shape[:2] != pf.shape[:2]:
        raise ValueError('heatmap and paf sizes differ: %r vs %r' % (hm.shape, pf.shape))
    H, W = hm.shape[:2]
    pk, pairs = 128, 32768
    while True:
        ctx = _pp._context(H, W, pk, pairs, 128)
        ctx.set_mode(m)
        ctx.load_fullres(hm[None], pf[None], 1, H, W)
        ctx.decode_fullres(thre1, thre2)
        try:
            res = Results(ctx.fetch(), m)
            break
        except _ffi.DecodeOverflow:
            if pk >= 4096:
                raise
            pk, pairs = pk * 4, pairs * 16
    cand, peaks, subset = res.image(0)
    if hm.dtype == np.float32:
        peaks = res.all_peaks(0, np.float32)
    return cand, peaks, subset


def _resize_context(h, w):
    # a context of the cached kind: the image-side entry points only use its stream and scratch
    return _pp._context(max(h, 1), max(w, 1), 16, 16, 16)


def resize_pad(img, scale, stride=STRIDE, pad_value=PAD_VALUE):
    """predict.py:56-57 on the device: ``cv2.resize(img, (0, 0), fx=scale, fy=scale, INTER_CUBIC)``
    then ``pad_right_down_corner``.  Returns ``(img_padded, pad)``; a uint8 image stays uint8
    (float32 evaluation, rounded half to even and saturated)."""
    img = np.asarray(img)
    if img.dtype not in (np.uint8, np.float32):
        img = img.astype(np.float32)
    ctx = _resize_context(img.shape[0], img.shape[1])
    return ctx.image_resize_pad(img, scale, stride, pad_value)


def detect(img, predict_func, draw_result=False, resize=None):
    """predict.py:45 ``detect(img, predict_func, draw_result=False)`` ->
    ``(candidate, all_peaks, subset)``.  The multi-scale loop is predict.py:49-72: the
    image-side cubic resize and padding (predict.py:56-57) run on the device
    (``resize_pad``) unless ``resize(img, scale)`` is given, in which case its result is
    padded on the host; the network call is the caller's ``predict_func``.  Everything behind
    the network calls (predict.py:63-254) is one enqueue on the process's cached context
    (``decode_scales``): no averaged map is stored, read back or uploaded.

    Parity of the image-side resize is unpinned: ``cv2.resize`` on 8-bit images is a
    fixed-point cubic whose last bit depends on the OpenCV build, ``resize_pad`` evaluates the
    cubic in float32 and rounds half to even, and the reference holds no fixture for either.
    Callers that have OpenCV and want its very bytes pass ``resize=lambda im, s: cv2.resize(im,
    (0, 0), fx=s, fy=s, interpolation=cv2.INTER_CUBIC)``."""
    h, w = img.shape[:2]
    outs, pads = [], []
    for x in SCALE_SEARCH:
        scale = x * IMG_Y / h                                   # predict.py:49
        if resize is None:
            padded, pad = resize_pad(img, scale, STRIDE, PAD_VALUE)
        else:
            padded, pad = pad_right_down_corner(resize(img, scale), STRIDE, PAD_VALUE)
        heatmap, paf = predict_func(np.expand_dims(padded, axis=0))   # predict.py:59-61
        outs.append((heatmap, paf))
        pads.append(pad)
    return decode_scales(outs, pads, (h, w), THRE1, THRE2, mode='D')


def render_lists(candidate, all_peaks, subset):
    """The draw lists of predict.py:265-299: ``circles (n, 3)`` = (x, y, colour) for every peak
    in part order, ``ellipses (m, 6)`` = (cx, cy, a, b, angle, colour), the arguments of
    ``cv2.ellipse2Poly`` at predict.py:297, for every limb (except 9 and 13) and person."""
    import math
    circles = [(int(p[0]), int(p[1]), i) for i in range(18) for p in all_peaks[i]]
    ellipses = []
    candidate = np.asarray(candidate, np.float64).reshape(-1, 4)
    for i in range(19):
        if i == 9 or i == 13:                                    # predict.py:285
            continue
        for n in range(len(subset)):
            index = np.asarray(subset[n])[np.array(_pp.limb_seq[i])]
            if -1 in index:
                continue
            Y = candidate[index.astype(int), 0]                  # predict.py:290-291 (the names are the
            X = candidate[index.astype(int), 1]                  # reference's: column 0 is x in mode D)
            mX, mY = np.mean(X), np.mean(Y)
            length = ((X[0] - X[1]) ** 2 + (Y[0] - Y[1]) ** 2) ** 0.5
            angle = math.degrees(math.atan2(X[0] - X[1], Y[0] - Y[1]))
            ellipses.append((int(mY), int(mX), int(length / 2), 4, int(angle), i % 18))
    return (np.array(circles, np.int32).reshape(-1, 3), np.array(ellipses, np.int32).reshape(-1, 6))


def visualize(img, candidate, all_peaks, subset, write=True):
    """predict.py:256-307 ``visualize(img, candidate, all_peaks, subset)``: the peak discs
    blended over the image (part.jpg) and the limb ellipses blended into the canvas
    (final.jpg), rasterised on the device.  Returns ``(part_image, final_image)``; with
    ``write`` and cv2 importable the two files are written as the reference does."""
    img = np.ascontiguousarray(img, np.uint8)
    circles, ellipses = render_lists(candidate, all_peaks, subset)
    ctx = _resize_context(img.shape[0], img.shape[1])
    part, final = ctx.render(img, circles, ellipses)
    if write:
        try:
            import cv2
        except ImportError:
            cv2 = None
        if cv2 is not None:
            cv2.imwrite('part.jpg', part)
            cv2.imwrite('final.jpg', final)
    return part, final
