"""Drop-in for the reference's ``utils/postprocess.py`` (same names, arguments and
return structures), computed by the HIP kernels of libposedecode.so.

    peaks = get_peaks(heatmap, threshold)                        # utils/postprocess.py:26
    conns, special = get_connections(peaks, pafs, threshold)     # utils/postprocess.py:78
    persons = multi_person_parse(peaks, conns, special)          # utils/postprocess.py:163

Known deviations, all deliberate:
* the stray ``print(connection, len(found))`` for limb 18 (utils/postprocess.py:224-225)
  is produced only when the module attribute ``debug_print`` is set to True;
* peak ids inside one part must be unique (they are when they come from get_peaks);
* a sample index outside the PAF array is clamped instead of raising / wrapping;
* at most 400 persons can be open at once while one image is parsed.
There is no CPU fallback: without the built library and a GPU these functions raise.
"""
import numpy as np

from . import _ffi
from .config import limb_seq, LIMB_FROM, LIMB_TO  # noqa: F401  (limb_seq is part of the reference module's surface)

_MODE = _ffi.MODE_P
debug_print = False   # True: multi_person_parse prints the reference's limb-18 debug lines
_state = {'ctx': None, 'key': None}


def _context(H, W, pk, pairs, persons, device_id=0):
    """One cached context, re-created when a capacity has to grow."""
    key = _state['key']
    if key is not None and key[0] == device_id and key[1] >= H and key[2] >= W and key[3] >= pk \
            and key[4] >= pairs and key[5] >= persons:
        return _state['ctx']
    if _state['ctx'] is not None:
        _state['ctx'].close()
        _state['ctx'] = None
    new_key = (device_id, max(H, key[1] if key else 0), max(W, key[2] if key else 0),
               max(pk, key[3] if key else 0), max(pairs, key[4] if key else 0),
               max(persons, key[5] if key else 0))
    _state['ctx'] = _ffi.Context(device_id=device_id, max_batch=1, max_h=new_key[1], max_w=new_key[2],
                                 max_peaks_per_part=new_key[3], max_persons=new_key[5],
                                 max_pairs_per_image=new_key[4])
    _state['key'] = new_key
    return _state['ctx']


def is_isolated(a, b):
    """utils/postprocess.py:14-16: no part index is assigned in both rows."""
    a = np.asarray(a)
    b = np.asarray(b)
    return not bool(np.any((a >= 0) & (b >= 0)))


def _id_to_idx(peaks, part_id):
    """utils/postprocess.py:18-23: index of the (last) peak whose id is part_id."""
    res = None
    for idx, item in enumerate(peaks):
        if item[3] == part_id:
            res = idx
    if res is None:
        raise UnboundLocalError("local variable 'res' referenced before assignment")
    return res


def _as_float_array(a, min_channels, name):
    a = np.asarray(a)
    if a.ndim != 3 or a.shape[2] < min_channels:
        raise ValueError('%s must be (H, W, >=%d), got %r' % (name, min_channels, a.shape))
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return np.ascontiguousarray(a)


def _peaks_from_packed(res, b, score_dtype, xy_order=False):
    """Packed peaks of image b -> the reference's list of 18 lists of tuples."""
    out = []
    counts = res['part_count'][b]
    start = int(res['peak_start'][b])
    pk = res['peaks']
    pid = 0
    for part in range(18):
        n = int(counts[part])
        seg = pk[start + pid:start + pid + n]
        ys = seg['y'].astype(np.int64)
        xs = seg['x'].astype(np.int64)
        sc = seg['score'].astype(score_dtype)
        if xy_order:
            out.append([(xs[i], ys[i], sc[i], pid + i) for i in range(n)])
        else:
            out.append([(ys[i], xs[i], sc[i], pid + i) for i in range(n)])
        pid += n
    return out


def _with_starts(res):
    """Per-image start offsets into the packed arrays."""
    pc = res['part_count'].sum(axis=1)
    res['peak_start'] = np.concatenate([[0], np.cumsum(pc)])
    lc = np.maximum(res['limb_count'], 0).sum(axis=1)
    res['conn_start'] = np.concatenate([[0], np.cumsum(lc)])
    res['person_start'] = np.concatenate([[0], np.cumsum(res['person_count'])])
    return res


def get_peaks(heatmap, threshold):
    """Peaks of the 18 part heatmaps: list of 18 lists of ``(y, x, score, id)``."""
    hm = _as_float_array(heatmap, 18, 'heatmap')
    H, W = hm.shape[:2]
    pk = 256
    while True:
        ctx = _context(H, W, pk, 1, 1)
        ctx.set_mode(_MODE)
        ctx.load_fullres(hm[None], None, 1, H, W)
        ctx.peaks(threshold)
        ctx.pack()
        try:
            res = _with_starts(ctx.fetch())
            break
        except _ffi.DecodeOverflow:
            if pk >= H * W:
                raise
            pk = min(H * W, pk * 8)
    return _peaks_from_packed(res, 0, hm.dtype)


def _upload_peaks(ctx, peaks):
    counts = [len(p) for p in peaks]
    flat = [t for plist in peaks for t in plist]
    y = np.array([int(t[0]) for t in flat], np.int32)
    x = np.array([int(t[1]) for t in flat], np.int32)
    sc = np.array([float(t[2]) for t in flat], np.float64)
    ids = np.array([int(t[3]) for t in flat], np.int32)
    ctx.set_peaks(counts, y, x, sc, ids)


def _pair_capacity(peaks):
    return max(1, sum(len(peaks[a]) * len(peaks[b]) for a, b in zip(LIMB_FROM, LIMB_TO)))


def get_connections(peaks, pafs, threshold):
    """``(all_connections, special_limb_idxs)``: per limb ``[]`` when one end has no
    peak, else an ``(n, 5)`` float64 array ``[id_a, id_b, score, conf_a, conf_b]``."""
    if len(peaks) < 18:
        raise ValueError('peaks must hold 18 lists')
    paf = _as_float_array(pafs, 38, 'pafs')
    H, W = paf.shape[:2]
    pk = max(16, max(len(p) for p in peaks))
    ctx = _context(H, W, pk, _pair_capacity(peaks), 1)
    ctx.set_mode(_MODE)
    ctx.load_fullres(None, paf[None], 1, H, W)
    _upload_peaks(ctx, peaks[:18])
    ctx.connections(threshold)
    ctx.pack()
    res = _with_starts(ctx.fetch())
    return _conns_from_packed(res, 0, peaks, _MODE)


def _conns_from_packed(res, b, peaks, mode):
    all_conn, special = [], []
    at = int(res['conn_start'][b])
    cn = res['conns']
    for k in range(19):
        n = int(res['limb_count'][b, k])
        if n < 0:
            special.append(k)
            all_conn.append([])
            continue
        seg = cn[at:at + n]
        rows = np.zeros((n, 5))
        rows[:, 0] = seg['id_a']
        rows[:, 1] = seg['id_b']
        rows[:, 2] = seg['score']
        if mode == _ffi.MODE_P:
            pa, pb = peaks[LIMB_FROM[k]], peaks[LIMB_TO[k]]
            rows[:, 3] = [pa[i][2] for i in seg['ia']]
            rows[:, 4] = [pb[j][2] for j in seg['ib']]
        else:
            rows[:, 3] = seg['ia']
            rows[:, 4] = seg['ib']
        all_conn.append(rows)
        at += n
    return all_conn, special


def _upload_connections(ctx, peaks, all_connections, special_limb_idx):
    counts, rows = [], []
    for k in range(19):
        if k in special_limb_idx:
            counts.append(-1)
            continue
        pa, pb = peaks[LIMB_FROM[k]], peaks[LIMB_TO[k]]
        conn = all_connections[k]
        counts.append(len(conn))
        for c in conn:
            id_a, id_b = int(c[0]), int(c[1])
            conf_a = pa[_id_to_idx(pa, id_a)][2]
            conf_b = pb[_id_to_idx(pb, id_b)][2]
            rows.append([id_a, id_b, float(c[2]), float(conf_a), float(conf_b)])
    ctx.set_connections(counts, np.array(rows, np.float64).reshape(-1, 5))
    return sum(max(c, 0) for c in counts)


def multi_person_parse(peaks, all_connections, special_limb_idx):
    """Person rows ``(n, 20)`` float64: 18 peak ids (-1 = missing), score, part count."""
    n_conn = sum(len(c) for k, c in enumerate(all_connections) if k not in special_limb_idx)
    persons_cap = int(min(400, max(128, n_conn + 1)))
    pk = max([16] + [len(c) for c in all_connections])
    ctx = _context(1, 1, pk, 1, persons_cap)
    ctx.set_mode(_MODE)
    _upload_connections(ctx, peaks, all_connections, special_limb_idx)
    want_print = debug_print and 18 not in special_limb_idx and len(all_connections[18]) > 0
    if want_print:
        ctx.debug_found(True)
    ctx.parse()
    ctx.pack()
    res = ctx.fetch()
    if want_print:
        found = ctx.debug_found(False, read=True)[0]
        for i, connection in enumerate(all_connections[18]):     # utils/postprocess.py:224-225
            print(connection, int(found[i]))
    return res['persons'].copy()
