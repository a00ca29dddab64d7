"""Seeded synthetic network outputs (19 heatmaps + 38 PAF channels at stride 8).

There is no dataset and no checkpoint here, so tests and the benchmark decode
synthetic scenes.  The label rules follow the reference's training labels so the
statistics resemble real network outputs:

* keypoint heatmaps: ``max_p exp(-d^2 / (2 * 7^2))`` sampled at cell centres
  ``3.5 + 8 g`` (reader.py:262-285, sigma = 7 from cfgs/config_vgg19.py:67),
  channel 18 = ``1 - max`` (reader.py:288-291);
* PAFs: the limb's unit vector on every cell within ``thre = 1`` cell of the
  segment inside its bounding box, averaged where limbs overlap
  (reader.py:293-342).  ``paf_order='xy'`` stores the x component in channel 2k
  (reader.py:334-335, what predict.py:140-152 reads); ``'yx'`` stores the y
  component first (what utils/postprocess.py:123-128 reads).

Gaussian noise N(0, 0.01) is added and the result cast to float32.  With
``quantize=True`` values are rounded to multiples of 2**-12 so fixtures stay
exactly representable and compact.
"""
import numpy as np

from .config import LIMB_FROM, LIMB_TO, STRIDE

# 18-joint upright template, (x, y) in a unit box: x in [-0.5, 0.5], y in [0, 1].
# Order: nose, neck, r-shoulder, r-elbow, r-wrist, l-shoulder, l-elbow, l-wrist,
# r-hip, r-knee, r-ankle, l-hip, l-knee, l-ankle, r-eye, l-eye, r-ear, l-ear.
_TEMPLATE = np.array([
    [0.00, 0.08], [0.00, 0.20], [-0.22, 0.21], [-0.30, 0.38], [-0.34, 0.54],
    [0.22, 0.21], [0.30, 0.38], [0.34, 0.54], [-0.13, 0.55], [-0.15, 0.77],
    [-0.16, 0.98], [0.13, 0.55], [0.15, 0.77], [0.16, 0.98], [-0.045, 0.05],
    [0.045, 0.05], [-0.10, 0.07], [0.10, 0.07]], dtype=np.float64)


def people(H, W, P, rng):
    """Joint coordinates (P, 18, 2) as (x, y) in full-resolution pixels."""
    out = np.zeros((P, 18, 2))
    for p in range(P):
        size = rng.uniform(0.35, 0.7) * H
        width = 0.6 * size
        cx = rng.uniform(0.15 * W, 0.85 * W)
        top = rng.uniform(0.02 * H, max(0.03 * H, 0.98 * H - size))
        jitter = rng.normal(0.0, 0.01 * size, size=(18, 2))
        out[p, :, 0] = cx + _TEMPLATE[:, 0] * width + jitter[:, 0]
        out[p, :, 1] = top + _TEMPLATE[:, 1] * size + jitter[:, 1]
    out[..., 0] = np.clip(out[..., 0], 2.0, W - 3.0)
    out[..., 1] = np.clip(out[..., 1], 2.0, H - 3.0)
    return out


def render(joints, h, w, paf_order='xy', noise=0.01, rng=None, quantize=False):
    """Render (h, w, 19) heatmaps and (h, w, 38) PAFs for joints given in pixels."""
    P = joints.shape[0]
    start = STRIDE / 2.0 - 0.5
    gy = start + STRIDE * np.arange(h)[:, None]
    gx = start + STRIDE * np.arange(w)[None, :]
    heat = np.zeros((h, w, 19))
    for i in range(18):
        for p in range(P):
            d2 = (gx - joints[p, i, 0]) ** 2 + (gy - joints[p, i, 1]) ** 2
            e = d2 / 2.0 / 7.0 / 7.0
            v = np.where(e > 4.6052, 0.0, np.minimum(np.exp(-e), 1.0))
            heat[:, :, i] = np.maximum(heat[:, :, i], v)
    heat[:, :, 18] = 1.0 - heat[:, :, :18].max(axis=2)

    paf = np.zeros((h, w, 38))
    cy = np.arange(h)[:, None].astype(np.float64)
    cx = np.arange(w)[None, :].astype(np.float64)
    for k in range(19):
        count = np.zeros((h, w))
        acc_x = np.zeros((h, w))
        acc_y = np.zeros((h, w))
        for p in range(P):
            fx, fy = joints[p, LIMB_FROM[k]] / STRIDE
            tx, ty = joints[p, LIMB_TO[k]] / STRIDE
            bx, by = tx - fx, ty - fy
            nrm = np.hypot(bx, by)
            if nrm < 1e-8:
                continue
            bx, by = bx / nrm, by / nrm
            x0 = int(max(round(min(fx, tx) - 1), 0)); x1 = int(min(round(max(fx, tx) + 1), w))
            y0 = int(max(round(min(fy, ty) - 1), 0)); y1 = int(min(round(max(fy, ty) + 1), h))
            if x1 <= x0 or y1 <= y0:
                continue
            dist = np.abs((cx[:, x0:x1] - fx) * by - (cy[y0:y1, :] - fy) * bx)
            m = dist <= 1.0
            acc_x[y0:y1, x0:x1] += np.where(m, bx, 0.0)
            acc_y[y0:y1, x0:x1] += np.where(m, by, 0.0)
            count[y0:y1, x0:x1] += m
        nz = count > 0
        acc_x[nz] /= count[nz]
        acc_y[nz] /= count[nz]
        if paf_order == 'xy':
            paf[:, :, 2 * k], paf[:, :, 2 * k + 1] = acc_x, acc_y
        else:
            paf[:, :, 2 * k], paf[:, :, 2 * k + 1] = acc_y, acc_x

    if noise and rng is not None:
        heat = heat + rng.normal(0.0, noise, size=heat.shape)
        paf = paf + rng.normal(0.0, noise, size=paf.shape)
    heat = heat.astype(np.float32)
    paf = paf.astype(np.float32)
    if quantize:
        heat = (np.round(heat * 4096.0) / 4096.0).astype(np.float32)
        paf = (np.round(paf * 4096.0) / 4096.0).astype(np.float32)
    return heat, paf


def scene(H, W, P, seed, paf_order='xy', noise=0.01, quantize=False, scale=1.0):
    """One synthetic image: network outputs for an H x W input holding P people.

    ``scale`` renders the same scene as the network would see it at another
    test-time scale (predict.py:49-57): the grid becomes
    ``ceil(H * scale / 8) x ceil(W * scale / 8)``.
    """
    rng = np.random.RandomState(seed)
    joints = people(H, W, P, rng)
    hs = -(-int(round(H * scale)) // STRIDE)
    ws = -(-int(round(W * scale)) // STRIDE)
    return render(joints * scale, hs, ws, paf_order, noise, rng, quantize)


def batch(B, H, W, people_per_image, seed0=0, paf_order='xy', noise=0.01, quantize=False):
    """(B, h, w, 19) and (B, h, w, 38) float32 arrays; ``people_per_image`` is an
    int or a (lo, hi) inclusive range drawn per image from its own seed."""
    h, w = -(-H // STRIDE), -(-W // STRIDE)
    heat = np.empty((B, h, w, 19), np.float32)
    paf = np.empty((B, h, w, 38), np.float32)
    for b in range(B):
        if isinstance(people_per_image, tuple):
            P = int(np.random.RandomState(100003 + seed0 + b).randint(
                people_per_image[0], people_per_image[1] + 1))
        else:
            P = int(people_per_image)
        heat[b], paf[b] = scene(H, W, P, seed0 + b, paf_order, noise, quantize)
    return heat, paf
