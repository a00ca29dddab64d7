"""CPU restatement (numpy) of the reference's OpenPose decode path.

TEST INFRASTRUCTURE ONLY.  This module is the checker for the HIP path: it may be
imported by ``tests/``, by ``__graft_entry__.smoke()`` and by the ``cpu_baseline``
leg of ``bench.py`` and by nothing else.  The product package never imports it.

Pinning status
--------------
* mode 'P' (utils/postprocess.py semantics: ``get_peaks``, ``get_connections``,
  ``multi_person_parse``) is PINNED: ``tools/make_golden.py`` ran the reference's
  own utils/postprocess.py in the build container on seeded scenes and the
  outputs are committed under ``tests/golden/``; ``tests/test_oracle.py`` checks
  this module against them.
* ``gaussian_filter_sigma3`` is PINNED against the installed
  ``scipy.ndimage.gaussian_filter`` (the reference's third-party dependency,
  version un-pinned upstream; 1.15.3 here) bit for bit.
* mode 'D' (predict.py detect() semantics) and ``resize_cubic`` (``cv2.resize``
  with INTER_CUBIC) are PARITY UNPINNED: predict.py imports tensorpack,
  tensorflow and cv2, none of which exist here, and the reference holds no test
  or fixture for them.  They follow the source text / OpenCV's published
  algorithm and are covered by hand-derived known-answer cases only.

Every function cites the reference lines it follows.  The arithmetic is written
so that each floating-point operation is the one the reference performs, in the
reference's order; where numpy vectorisation is used it only batches independent
elementwise IEEE operations and never reassociates a sum.
"""
import math

import numpy as np

LIMB_FROM = (1, 8, 9, 1, 11, 12, 1, 2, 3, 2, 1, 5, 6, 5, 1, 0, 0, 14, 15)   # cfgs/config_vgg19.py:56
LIMB_TO = (8, 9, 10, 11, 12, 13, 2, 3, 4, 16, 5, 6, 7, 17, 0, 14, 15, 16, 17)  # cfgs/config_vgg19.py:57
MID_NUM = 10   # utils/postprocess.py:79, predict.py:123


# ----------------------------------------------------------------------------
# scipy.ndimage.gaussian_filter(x, sigma=3)  (utils/postprocess.py:44, predict.py:93)
# ----------------------------------------------------------------------------

def gaussian_weights(sigma=3.0, truncate=4.0):
    """scipy _filters.py:_gaussian_kernel1d with order 0 and radius int(truncate*sigma+0.5)."""
    radius = int(truncate * float(sigma) + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x ** 2)
    return phi / phi.sum()


def _reflect_index(i, n):
    """scipy mode='reflect' (d c b a | a b c d | d c b a), valid for any offset."""
    period = 2 * n
    i = np.mod(i, period)
    return np.where(i >= n, period - 1 - i, i)


def _correlate_axis0(x, w):
    """Symmetric-kernel branch of scipy's C correlate1d along axis 0, in double:
    t = x[c]*w[r]; for j = r..1 (outermost first): t += (x[c-j] + x[c+j]) * w[r-j]."""
    n = x.shape[0]
    r = len(w) // 2
    xd = x.astype(np.float64)
    rows = np.arange(n)
    out = xd * w[r]
    for j in range(r, 0, -1):
        lo = xd[_reflect_index(rows - j, n)]
        hi = xd[_reflect_index(rows + j, n)]
        out = out + (lo + hi) * w[r - j]
    return out


def gaussian_filter_sigma3(x):
    """Restated scipy.ndimage.gaussian_filter(x, sigma=3): axis 0 then axis 1,
    intermediate and result stored in x's dtype."""
    w = gaussian_weights(3.0)
    x = np.asarray(x)
    dt = x.dtype if x.dtype in (np.float32, np.float64) else np.float64
    t = _correlate_axis0(x, w).astype(dt)
    t = _correlate_axis0(t.T, w).T.astype(dt)
    return np.ascontiguousarray(t)


# ----------------------------------------------------------------------------
# get_peaks  (utils/postprocess.py:26-75; twin predict.py:88-111)
# ----------------------------------------------------------------------------

def get_peaks(heatmap, threshold, mode='P', use_scipy=False):
    """18 lists of (y, x, score, id) [mode P] or (x, y, score, id) [mode D]."""
    heatmap = np.asarray(heatmap)
    if use_scipy:
        from scipy.ndimage import gaussian_filter
    out, counter = [], 0
    for part in range(18):
        ori = heatmap[:, :, part]
        sm = gaussian_filter(ori, sigma=3) if use_scipy else gaussian_filter_sigma3(ori)
        # neighbours along y (postprocess.py:50-53) and x (:54-57); outside = 0.0
        yprev = np.zeros(sm.shape); yprev[1:, :] = sm[:-1, :]
        ynext = np.zeros(sm.shape); ynext[:-1, :] = sm[1:, :]
        xprev = np.zeros(sm.shape); xprev[:, 1:] = sm[:, :-1]
        xnext = np.zeros(sm.shape); xnext[:, :-1] = sm[:, 1:]
        keep = (sm >= yprev) & (sm >= ynext) & (sm >= xprev) & (sm >= xnext) & (sm > threshold)
        ys, xs = np.nonzero(keep)                       # row-major: y asc, then x asc
        plist = []
        for n in range(len(ys)):
            y, x = ys[n], xs[n]
            first, second = (y, x) if mode == 'P' else (x, y)
            plist.append((first, second, ori[y, x], counter + n))
        out.append(plist)
        counter += len(ys)
    return out


# ----------------------------------------------------------------------------
# get_connections  (utils/postprocess.py:78-160; twin predict.py:120-174)
# ----------------------------------------------------------------------------

def _linspace10(a, b):
    """np.linspace(a, b, 10) for scalar endpoints, vectorised over pairs:
    i*((b-a)/9) + a with the last point forced to b; a when the step is 0."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    step = (b - a) / 9.0
    i = np.arange(MID_NUM, dtype=np.float64)[None, :]
    y = i * step[:, None] + a[:, None]
    y[:, -1] = b
    return y


def pair_scores(ya, xa, yb, xb, paf2, img_h, threshold, mode='P'):
    """Score every (a, b) pair of one limb.  ya.. are 1-D arrays of equal length
    (already expanded a-major, b-minor); paf2 = pafs[:, :, 2k:2k+2].
    Returns (score, ok) arrays."""
    ya = np.asarray(ya, np.int64); xa = np.asarray(xa, np.int64)
    yb = np.asarray(yb, np.int64); xb = np.asarray(xb, np.int64)
    dy = yb - ya
    dx = xb - xa
    d2 = dy * dy + dx * dx
    if mode == 'P':
        length = np.power(d2, 0.5)                     # postprocess.py:108 (numpy pow)
        norm = np.sqrt(d2.astype(np.float64))          # np.linalg.norm, postprocess.py:110
        skip = length == 0                             # postprocess.py:112
        with np.errstate(divide='ignore', invalid='ignore'):
            u_y = dy / length
            u_x = dx / length
    else:
        norm = np.maximum(np.sqrt(d2.astype(np.float64)), 1e-5)   # predict.py:141
        skip = np.zeros(len(d2), bool)
        u_y = dy / norm
        u_x = dx / norm
        length = norm
    ly = _linspace10(ya, yb)
    lx = _linspace10(xa, xb)
    if mode == 'P':
        iy = ly.astype(np.int64)                       # int(y): truncation, postprocess.py:123
        ix = lx.astype(np.int64)
    else:
        iy = np.rint(ly).astype(np.int64)              # int(round(np.float64)): half-even, predict.py:147
        ix = np.rint(lx).astype(np.int64)
    v0 = paf2[iy, ix, 0].astype(np.float64)
    v1 = paf2[iy, ix, 1].astype(np.float64)
    if mode == 'P':
        # channel 2k pairs with dy, 2k+1 with dx  (postprocess.py:123-128)
        mid = v0 * u_y[:, None] + v1 * u_x[:, None]
    else:
        # channel 2k pairs with dx, 2k+1 with dy  (predict.py:147-152)
        mid = v0 * u_x[:, None] + v1 * u_y[:, None]
    acc = 0 + mid[:, 0]                                # builtin sum(): left to right from 0
    for i in range(1, MID_NUM):
        acc = acc + mid[:, i]
    with np.errstate(divide='ignore', invalid='ignore'):
        prior = np.minimum(0.5 * img_h / norm - 1, 0)  # postprocess.py:129, predict.py:154
        score = acc / MID_NUM + prior
    c1 = (mid > threshold).sum(axis=1) > 0.8 * MID_NUM
    with np.errstate(invalid='ignore'):
        ok = c1 & (score > 0) & ~skip
    return score, ok


def get_connections(peaks, pafs, threshold, mode='P'):
    """(all_connections, special_limb_idxs).  Rows: mode P [id_a, id_b, score,
    conf_a, conf_b]; mode D [id_a, id_b, score, i, j]."""
    pafs = np.asarray(pafs)
    img_h = pafs.shape[0]
    all_conn, special = [], []
    for k in range(19):
        ca, cb = peaks[LIMB_FROM[k]], peaks[LIMB_TO[k]]
        na, nb = len(ca), len(cb)
        if na * nb == 0:
            special.append(k)
            all_conn.append([])
            continue
        ia = np.repeat(np.arange(na), nb)
        ib = np.tile(np.arange(nb), na)
        if mode == 'P':
            ya = np.array([p[0] for p in ca]); xa = np.array([p[1] for p in ca])
            yb = np.array([p[0] for p in cb]); xb = np.array([p[1] for p in cb])
        else:
            xa = np.array([p[0] for p in ca]); ya = np.array([p[1] for p in ca])
            xb = np.array([p[0] for p in cb]); yb = np.array([p[1] for p in cb])
        score, ok = pair_scores(ya[ia], xa[ia], yb[ib], xb[ib],
                                pafs[:, :, 2 * k:2 * k + 2], img_h, threshold, mode)
        cand = np.nonzero(ok)[0]
        # stable descending sort on score (list.sort(reverse=True) keeps ties in order)
        order = cand[np.argsort(-score[cand], kind='stable')]
        used_a, used_b, rows = set(), set(), []
        for c in order:
            i, j = int(ia[c]), int(ib[c])
            if i in used_a or j in used_b:
                continue
            used_a.add(i); used_b.add(j)
            if mode == 'P':
                rows.append([ca[i][3], cb[j][3], score[c], ca[i][2], cb[j][2]])
            else:
                rows.append([ca[i][3], cb[j][3], score[c], i, j])
            if len(rows) >= min(na, nb):
                break
        all_conn.append(np.array(rows, dtype=np.float64).reshape(-1, 5))
    return all_conn, special


# ----------------------------------------------------------------------------
# multi_person_parse  (utils/postprocess.py:163-237; twin predict.py:179-252)
# ----------------------------------------------------------------------------

class TooManyOwners(IndexError):
    """predict.py:195-200 writes subset_idx[found] with found == 2 -> IndexError."""


def multi_person_parse(peaks, all_connections, special_limb_idx, mode='P'):
    conf = {}
    for plist in peaks:
        for p in plist:
            conf[int(p[3])] = p[2]          # ids are unique (postprocess.py:68)
    persons = []                            # list of 20-vectors, order = row order
    for k in range(19):
        if k in special_limb_idx:
            continue
        a, b = LIMB_FROM[k], LIMB_TO[k]
        for row in all_connections[k]:
            id_a, id_b = int(row[0]), int(row[1])
            s = row[2]
            found = [n for n, p in enumerate(persons) if p[a] == id_a or p[b] == id_b]
            if mode == 'D' and len(found) > 2:
                raise TooManyOwners(k)
            if len(found) == 0:
                if mode == 'D' or k < 17:                       # postprocess.py:196 / predict.py:228
                    p = -1 * np.ones(20)
                    p[a], p[b] = id_a, id_b
                    p[19] = 2
                    if mode == 'P':
                        p[18] = conf[id_a] + conf[id_b] + s     # postprocess.py:203
                    else:
                        p[18] = (0 + conf[id_a] + conf[id_b]) + s   # predict.py:243 (sum() then +)
                    persons.append(p)
            elif len(found) == 1:
                p = persons[found[0]]
                if p[b] != id_b:                                # postprocess.py:210 / predict.py:205
                    p[b] = id_b
                    p[19] += 1
                    p[18] += conf[id_b] + s
            elif len(found) == 2:
                p0, p1 = persons[found[0]], persons[found[1]]
                overlap = np.any((p0[:18] >= 0) & (p1[:18] >= 0))
                if not overlap:
                    if mode == 'P':
                        p0[:18] = (p0[:18] + 1) + p1[:18]       # postprocess.py:219
                    else:
                        p0[:18] += (p1[:18] + 1)                # predict.py:216
                    p0[18:] += p1[18:]
                    p0[18] += s
                    del persons[found[1]]
                elif mode == 'P' or p0[b] != id_b:              # postprocess.py:210-214 / predict.py:222
                    p0[b] = id_b
                    p0[19] += 1
                    p0[18] += conf[id_b] + s
            # mode P, len(found) >= 3: nothing (postprocess.py:196-222 has no branch)
    keep = [p for p in persons if not (p[19] < 4 or p[18] / p[19] < 0.4)]   # :228-232
    return np.array(keep, dtype=np.float64).reshape(-1, 20)


# ----------------------------------------------------------------------------
# cv2.resize(..., INTER_CUBIC) restated  (predict.py:56,63,65,67,69)  -- UNPINNED
# ----------------------------------------------------------------------------

def cubic_table(src_len, dst_len, scale=None):
    """Per destination index: first tap (floor(s) - 1) and the 4 float32 weights.
    OpenCV: s = (d + 0.5) * scale - 0.5 in double, cast to float; Keys kernel with
    A = -0.75 evaluated in float32 (interpolateCubic)."""
    if scale is None:
        scale = 1.0 / (float(dst_len) / float(src_len))
    d = np.arange(dst_len, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    t = (f - s.astype(np.float32)).astype(np.float32)
    A = np.float32(-0.75)
    one = np.float32(1.0)
    c0 = ((A * (t + one) - np.float32(5) * A) * (t + one) + np.float32(8) * A) * (t + one) - np.float32(4) * A
    c1 = ((A + np.float32(2)) * t - (A + np.float32(3))) * t * t + one
    u = one - t
    c2 = ((A + np.float32(2)) * u - (A + np.float32(3))) * u * u + one
    c3 = one - c0 - c1 - c2
    coef = np.stack([c0, c1, c2, c3], axis=1).astype(np.float32)
    return (s - 1).astype(np.int32), coef


def resize_cubic(src, dst_hw=None, fx=None, fy=None):
    """src (Hs, Ws, C) float32 -> (Hd, Wd, C) float32.  Horizontal pass then
    vertical pass, float32, taps clamped to the image, each 4-tap sum evaluated
    left to right ((s0*c0 + s1*c1) + s2*c2) + s3*c3 without fused multiply-add.
    Same size -> copy."""
    src = np.asarray(src, np.float32)
    Hs, Ws = src.shape[:2]
    if dst_hw is None:
        Hd, Wd = int(round(Hs * fy)), int(round(Ws * fx))
        sx, sy = 1.0 / fx, 1.0 / fy
    else:
        Hd, Wd = dst_hw
        sx = sy = None
    if (Hd, Wd) == (Hs, Ws):
        return src.copy()
    x0, cx = cubic_table(Ws, Wd, sx)
    y0, cy = cubic_table(Hs, Hd, sy)
    hor = None
    for j in range(4):
        term = src[:, np.clip(x0 + j, 0, Ws - 1), :] * cx[None, :, j, None]
        hor = term if hor is None else hor + term
    out = None
    for k in range(4):
        term = hor[np.clip(y0 + k, 0, Hs - 1), :, :] * cy[:, None, k, None]
        out = term if out is None else out + term
    return out.astype(np.float32)


def pad_amounts(h, w, stride=8):
    """predict.py:19-24: rows added below, columns added right."""
    return (stride - h % stride) % stride, (stride - w % stride) % stride


def upsample_average(heat_scales, paf_scales, pads, out_hw, stride=8):
    """predict.py:51-72.  heat_scales[m] is (hs, ws, 19) float32 network output of
    scale m, pads[m] = (down, right) padding that was added before the network."""
    H, W = out_hw
    n = len(heat_scales)
    heat_avg = np.zeros((H, W, heat_scales[0].shape[2]))
    paf_avg = np.zeros((H, W, paf_scales[0].shape[2]))
    for m in range(n):
        for src, avg in ((heat_scales[m], heat_avg), (paf_scales[m], paf_avg)):
            up = resize_cubic(src, fx=stride, fy=stride)                       # predict.py:63
            up = up[:up.shape[0] - pads[m][0], :up.shape[1] - pads[m][1], :]   # :64
            up = resize_cubic(np.ascontiguousarray(up), dst_hw=(H, W))         # :65
            avg += up / np.float32(n)                                          # :71 (f32 /, f64 +)
    return heat_avg, paf_avg


def decode(heat_avg, paf_avg, thre1=0.1, thre2=0.05, mode='D'):
    """predict.py:82-254 -> (candidate, all_peaks, subset)."""
    peaks = get_peaks(heat_avg, thre1, mode)
    conns, special = get_connections(peaks, paf_avg, thre2, mode)
    subset = multi_person_parse(peaks, conns, special, mode)
    flat = [item for plist in peaks for item in plist]
    candidate = np.array(flat, dtype=np.float64).reshape(-1, 4)
    return candidate, peaks, subset


def coco_records(candidate, subset, image_id, coco_to_ours):
    """evaluate.py:44-74: one result dict per person."""
    out = []
    for person in subset:
        kp = np.zeros((len(coco_to_ours), 3))
        for ci, oi in enumerate(coco_to_ours):
            pid = int(person[oi])
            if pid == -1:
                continue
            kp[ci, 0] = candidate[pid][0]
            kp[ci, 1] = candidate[pid][1]
            kp[ci, 2] = 1
        out.append({"image_id": image_id, "category_id": 1,
                    "keypoints": list(kp.reshape(-1)), "score": person[-2]})
    return out


# ----------------------------------------------------------------------------
# image side of the scale loop (predict.py:56-57)  -- UNPINNED (cv2 absent)
# ----------------------------------------------------------------------------

def image_resize_pad(img, scale, stride=8, pad_value=128, out_dtype=None):
    """cv2.resize(img, (0, 0), fx=scale, fy=scale, INTER_CUBIC) evaluated in float32 (the
    restatement above), uint8 results rounded half to even and saturated; then
    pad_right_down_corner (predict.py:19-32).  Returns (img_padded, pad)."""
    img = np.asarray(img)
    out_dtype = np.dtype(out_dtype or img.dtype)
    res = resize_cubic(img.astype(np.float32), fx=scale, fy=scale)
    down, right = pad_amounts(res.shape[0], res.shape[1], stride)
    out = np.full((res.shape[0] + down, res.shape[1] + right, res.shape[2]), np.float32(pad_value), np.float32)
    out[:res.shape[0], :res.shape[1]] = res
    if out_dtype == np.uint8:
        out = np.clip(np.rint(out), 0, 255).astype(np.uint8)
    return out, [0, 0, down, right]


# ----------------------------------------------------------------------------
# visualize (predict.py:256-307)  -- UNPINNED: cv2 is absent and the reference holds no image
# of its output.  The shapes follow OpenCV's published drawing algorithms (imgproc/drawing.cpp)
# as restated here from their description: the integer midpoint circle, ellipse2Poly's table of
# sines of whole degrees with rounded vertices, fillConvexPoly's outline + fixed-point scan.
# Lines are not clipped to the image first (cv2 moves the end points of an edge that crosses
# the border onto it before it walks the edge), so shapes cut by the border may differ there.
# ----------------------------------------------------------------------------

COLORS = np.array([[255, 0, 0], [255, 85, 0], [255, 170, 0], [255, 255, 0], [170, 255, 0], [85, 255, 0], [0, 255, 0],
                   [0, 255, 85], [0, 255, 170], [0, 255, 255], [0, 170, 255], [0, 85, 255], [0, 0, 255], [85, 0, 255],
                   [170, 0, 255], [255, 0, 255], [255, 0, 170], [255, 0, 85]], np.float32)    # predict.py:258-260


def _add_weighted(a, wa, b, wb):
    """cv2.addWeighted on 8-bit images: float32 arithmetic, rounded half to even, saturated."""
    return np.clip(np.rint(a * np.float32(wa) + b * np.float32(wb)), 0, 255).astype(np.float32)


def circle_rows(radius):
    """cv2.circle(..., thickness=-1) with line type 8 and no shift: the integer midpoint walk;
    half-width of the filled row at distance dy from the centre, for dy = 0 .. radius."""
    half = [-1] * (radius + 1)
    err, dx, dy, plus, minus = 0, radius, 0, 1, (radius << 1) - 1
    while dx >= dy:
        half[dy] = max(half[dy], dx)        # rows centre.y -+ dy span centre.x -+ dx
        half[dx] = max(half[dx], dy)        # rows centre.y -+ dx span centre.x -+ dy
        dy += 1
        err += plus
        plus += 2
        if err > 0:
            err -= minus
            dx -= 1
            minus -= 2
    return half


_SIN_TABLE = np.array([round(np.sin(np.deg2rad(d)), 7) for d in range(451)], np.float64).astype(np.float32)


def ellipse2poly(center, axes, angle, arc_start=0, arc_end=360, delta=1):
    """cv2.ellipse2Poly with integer centre and axes: vertices at every delta degrees from a
    float32 table of sines of whole degrees (7 decimals), evaluated in double, rounded half to
    even, consecutive duplicates dropped.  Returns a list of (x, y)."""
    while angle < 0:
        angle += 360
    while angle > 360:
        angle -= 360
    if arc_start > arc_end:
        arc_start, arc_end = arc_end, arc_start
    while arc_start < 0:
        arc_start += 360
        arc_end += 360
    while arc_end > 360:
        arc_end -= 360
        arc_start -= 360
    if arc_end - arc_start > 360:
        arc_start, arc_end = 0, 360
    alpha = float(_SIN_TABLE[450 - angle])      # cosine
    beta = float(_SIN_TABLE[angle])
    pts, prev = [], None
    i = arc_start
    while i < arc_end + delta:
        ang = min(i, arc_end)
        if ang < 0:
            ang += 360
        x = float(axes[0]) * float(_SIN_TABLE[450 - ang])
        y = float(axes[1]) * float(_SIN_TABLE[ang])
        pt = (int(np.rint(center[0] + x * alpha - y * beta)), int(np.rint(center[1] + x * beta + y * alpha)))
        if pt != prev:
            pts.append(pt)
            prev = pt
        i += delta
    if len(pts) == 1:
        pts = [tuple(center), tuple(center)]
    return pts


def line_pixels(p1, p2):
    """The 8-connected line cv2 draws between two pixels: walked from the left end, one step
    along the longer axis per pixel, a step along the shorter one whenever the error term is
    negative."""
    (x1, y1), (x2, y2) = p1, p2
    dx, dy = x2 - x1, y2 - y1
    if dx < 0:
        dx, dy, x1, y1 = -dx, -dy, x2, y2
    sy = -1 if dy < 0 else 1
    dy = abs(dy)
    steep = dy > dx
    major, minor = (dy, dx) if steep else (dx, dy)
    err = major - 2 * minor
    x, y, out = x1, y1, []
    for _ in range(major + 1):
        out.append((x, y))
        neg = err < 0
        err += -2 * minor + (2 * major if neg else 0)
        if steep:
            y += sy
            x += 1 if neg else 0
        else:
            x += 1
            y += sy if neg else 0
    return out


def _trunc_div(a, b):
    q = abs(a) // abs(b)
    return q if (a >= 0) == (b >= 0) else -q


def fill_convex_poly_mask(H, W, pts):
    """cv2.fillConvexPoly(img, pts, colour) with line type 8 and no shift, as a boolean mask:
    every edge drawn as a line, then rows ymin .. ymax filled between the two edge chains that
    leave the top vertex, x kept in 16.16 fixed point and advanced by a rounded slope per row."""
    XY_SHIFT = 16
    XY_ONE = 1 << XY_SHIFT
    mask = np.zeros((H, W), bool)
    n = len(pts)
    p0 = pts[-1]
    for p in pts:
        for (x, y) in line_pixels(p0, p):
            if 0 <= x < W and 0 <= y < H:
                mask[y, x] = True
        p0 = p
    xs_all = [p[0] for p in pts]
    ys_all = [p[1] for p in pts]
    ymin, ymax, xmin, xmax = min(ys_all), max(ys_all), min(xs_all), max(xs_all)
    imin = ys_all.index(ymin)                      # the first vertex of the lowest y
    if n < 3 or xmax < 0 or ymax < 0 or xmin >= W or ymin >= H:
        return mask
    ymax = min(ymax, H - 1)
    edge = [dict(idx=imin, di=1, x=-XY_ONE, dx=0, ye=ymin), dict(idx=imin, di=n - 1, x=-XY_ONE, dx=0, ye=ymin)]
    edges = n
    y = ymin
    while True:
        for e in edge:
            if y >= e['ye']:
                idx0, di = e['idx'], e['di']
                idx = idx0 + di
                if idx >= n:
                    idx -= n
                while True:
                    edges -= 1
                    if edges < 0:
                        break
                    ty = pts[idx][1]
                    if ty > y:
                        xs, xe = pts[idx0][0] << XY_SHIFT, pts[idx][0] << XY_SHIFT
                        e['ye'] = ty
                        e['dx'] = _trunc_div((xe - xs) * 2 + (ty - y), 2 * (ty - y))
                        e['x'] = xs
                        e['idx'] = idx
                        break
                    idx0 = idx
                    idx += di
                    if idx >= n:
                        idx -= n
        if edges < 0:
            break
        if y >= 0:
            left, right = (1, 0) if edge[0]['x'] > edge[1]['x'] else (0, 1)
            xx1 = (edge[left]['x'] + (XY_ONE >> 1)) >> XY_SHIFT
            xx2 = (edge[right]['x'] + (XY_ONE >> 1)) >> XY_SHIFT
            if xx2 >= 0 and xx1 < W:
                mask[y, max(xx1, 0):min(xx2, W - 1) + 1] = True
        edge[0]['x'] += edge[0]['dx']
        edge[1]['x'] += edge[1]['dx']
        y += 1
        if y > ymax:
            break
    return mask


def limb_ellipse(candidate, index):
    """predict.py:292-298: (centre, axes, angle) handed to cv2.ellipse2Poly for the limb between
    the two candidates."""
    import math
    Y = candidate[index.astype(int), 0]
    X = candidate[index.astype(int), 1]
    mX, mY = np.mean(X), np.mean(Y)
    length = ((X[0] - X[1]) ** 2 + (Y[0] - Y[1]) ** 2) ** 0.5
    angle = math.degrees(math.atan2(X[0] - X[1], Y[0] - Y[1]))
    return (int(mY), int(mX)), (int(length / 2), 4), int(angle)


def visualize(img, candidate, all_peaks, subset, limb_seq=None):
    """predict.py:256-307.  Returns (part.jpg image, final.jpg image)."""
    if limb_seq is None:
        limb_seq = list(zip(LIMB_FROM, LIMB_TO))
    img = np.asarray(img, np.uint8)
    H, W = img.shape[:2]
    yy, xx = np.mgrid[0:H, 0:W]
    base = img.astype(np.float32)
    canvas = base.copy()
    half = np.array(circle_rows(4) + [-1])
    for i in range(18):
        for p in all_peaks[i]:
            dx, dy = np.abs(xx - int(p[0])), np.minimum(np.abs(yy - int(p[1])), 5)      # predict.py:269
            canvas[dx <= half[dy]] = COLORS[i]
    part = _add_weighted(base, 0.3, canvas, 0.7)                            # predict.py:277
    candidate = np.asarray(candidate, np.float64).reshape(-1, 4)
    for i in range(19):
        if i == 9 or i == 13:
            continue
        for n in range(len(subset)):
            index = np.asarray(subset[n])[np.array(limb_seq[i])]
            if -1 in index:
                continue
            center, axes, ang = limb_ellipse(candidate, index)
            inside = fill_convex_poly_mask(H, W, ellipse2poly(center, axes, ang))     # predict.py:298-299
            canvas[inside] = _add_weighted(canvas[inside], 0.4, COLORS[i % 18][None, :], 0.6)   # predict.py:300
    return part.astype(np.uint8), canvas.astype(np.uint8)
