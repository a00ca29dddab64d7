"""The numpy oracle against the golden vectors produced by the reference's own
utils/postprocess.py (tools/make_golden.py), and against the installed scipy."""
import hashlib

import numpy as np
import pytest

from oracle import decode_ref as ref
from tests import golden_io as gio


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.mark.parametrize('name', gio.SCENES)
def test_scene_pipeline_matches_reference(name):
    g = gio.load(name)
    heat, paf = gio.scene_inputs(g)
    heat_full = ref.resize_cubic(heat, fx=8, fy=8).astype(np.float64)
    paf_full = ref.resize_cubic(paf, fx=8, fy=8).astype(np.float64)
    # the reference consumed exactly these full-resolution arrays
    assert _sha(heat_full) == str(g['heat_full_sha'])
    assert _sha(paf_full) == str(g['paf_full_sha'])
    want_peaks = gio.peaks_from(g['peaks'], g['peaks_part'])
    peaks = ref.get_peaks(heat_full, 0.1, 'P')
    gio.assert_peaks_equal(peaks, want_peaks, exact_score=True)
    conns, special = ref.get_connections(peaks, paf_full, 0.05, 'P')
    assert special == list(g['special'])
    gio.assert_conns_equal(conns, gio.conns_from(g['conns'], g['conns_limb'], g['conns_is_list']),
                           score_rtol=1e-12)
    persons = ref.multi_person_parse(peaks, conns, special, 'P')
    gio.assert_persons_equal(persons, g['persons'], score_rtol=1e-12)


@pytest.mark.parametrize('name', gio.PEAK_CASES)
def test_get_peaks_cases(name):
    g = gio.load('peaks_' + name)
    hm = (g['heatmap_q'].astype(np.float64) / 16384.0).astype(str(g['dtype']))
    peaks = ref.get_peaks(hm, float(g['threshold']), 'P')
    gio.assert_peaks_equal(peaks, gio.peaks_from(g['peaks'], g['peaks_part']), exact_score=True)
    if 'smoothed' in g.files:
        for c in range(18):
            got = ref.gaussian_filter_sigma3(hm[:, :, c])
            assert got.dtype == g['smoothed'].dtype
            np.testing.assert_array_equal(got, g['smoothed'][:, :, c])


@pytest.mark.parametrize('shape,dtype', [((40, 50), np.float64), ((40, 50), np.float32),
                                         ((5, 7), np.float64), ((368, 368), np.float64)])
def test_gaussian_restatement_is_bitwise_scipy(shape, dtype):
    from scipy.ndimage import gaussian_filter
    x = np.random.RandomState(3).randn(*shape).astype(dtype)
    np.testing.assert_array_equal(ref.gaussian_filter_sigma3(x), gaussian_filter(x, sigma=3))


def test_gaussian_weights_match_scipy():
    from scipy.ndimage import _filters
    np.testing.assert_array_equal(ref.gaussian_weights(3.0), _filters._gaussian_kernel1d(3, 0, 12))


@pytest.mark.parametrize('idx', gio.CONN_CASES)
def test_get_connections_cases(idx):
    g = gio.load('conn_case%d' % idx)
    paf = (g['paf_q'].astype(np.float64) / 4096.0).astype(str(g['dtype']))
    peaks = gio.peaks_from(g['peaks'], g['peaks_part'])
    conns, special = ref.get_connections(peaks, paf, float(g['threshold']), 'P')
    assert special == list(g['special'])
    gio.assert_conns_equal(conns, gio.conns_from(g['conns'], g['conns_limb'], g['conns_is_list']),
                           score_rtol=1e-12)


def test_multi_person_parse_cases():
    g = gio.load('parse_cases')
    for i in range(int(g['n'])):
        pre = '%03d_' % i
        peaks = gio.peaks_from(g[pre + 'peaks'], g[pre + 'peaks_part'])
        special = list(g[pre + 'special'])
        conns = gio.conns_from(g[pre + 'conns'], g[pre + 'conns_limb'])
        for k in special:
            conns[k] = []
        persons = ref.multi_person_parse(peaks, conns, special, 'P')
        gio.assert_persons_equal(persons, g[pre + 'persons'], score_rtol=1e-12)


def test_linspace_restatement():
    rng = np.random.RandomState(0)
    a = rng.randint(0, 700, 3000); b = rng.randint(0, 700, 3000)
    b[:50] = a[:50]
    got = ref._linspace10(a, b)
    for i in range(len(a)):
        np.testing.assert_array_equal(got[i], np.linspace(np.int64(a[i]), np.int64(b[i]), num=10))


def test_drawing_restatements_known_answers():
    """The shapes visualize() draws (predict.py:269, 298-299), restated from OpenCV's published
    algorithms; cv2 itself is not available, so these are worked by hand from the algorithms."""
    # midpoint circle of radius 4, filled: rows 9, 7, 7, 5, 1 pixels wide from the centre outwards
    assert ref.circle_rows(4) == [4, 3, 3, 2, 0]
    assert ref.circle_rows(1) == [1, 0]                     # a plus sign
    # the 8-connected line from (0, 0) to (5, 2): error term 1, -3, 3, -1, 5, 1
    assert ref.line_pixels((0, 0), (5, 2)) == [(0, 0), (1, 0), (2, 1), (3, 1), (4, 2), (5, 2)]
    assert ref.line_pixels((5, 2), (0, 0)) == ref.line_pixels((0, 0), (5, 2))        # walked from the left end
    assert ref.line_pixels((2, 7), (2, 3)) == [(2, 7), (2, 6), (2, 5), (2, 4), (2, 3)]
    for p1, p2 in [((3, 4), (40, 17)), ((10, 50), (12, -8)), ((0, 0), (0, 0)), ((9, 9), (-7, 25))]:
        px = ref.line_pixels(p1, p2)
        assert len(px) == max(abs(p1[0] - p2[0]), abs(p1[1] - p2[1])) + 1
        assert {px[0], px[-1]} == {p1, p2}
        assert all(max(abs(a[0] - b[0]), abs(a[1] - b[1])) == 1 for a, b in zip(px, px[1:]))
    # ellipse2Poly: closed, on the ellipse to within rounding, first vertex on the major axis
    pts = ref.ellipse2poly((50, 40), (20, 4), 30)
    assert pts[0] == (67, 50) and pts[0] == pts[-1]
    co, si = np.cos(np.deg2rad(30)), np.sin(np.deg2rad(30))
    for (x, y) in pts:
        u = ((x - 50) * co + (y - 40) * si) / 20.0
        v = ((y - 40) * co - (x - 50) * si) / 4.0
        assert abs(np.hypot(u, v) - 1) < 0.2
    assert ref.ellipse2poly((7, 9), (0, 0), 123) == [(7, 9), (7, 9)]
    assert ref.ellipse2poly((5, 5), (3, 1), -90) == ref.ellipse2poly((5, 5), (3, 1), 270)
    # fillConvexPoly: an axis-parallel rectangle fills exactly; an ellipse lies between the
    # analytic ellipse shrunk and grown by a pixel; a polygon of no area is its outline
    m = ref.fill_convex_poly_mask(20, 30, [(4, 3), (12, 3), (12, 9), (4, 9)])
    want = np.zeros((20, 30), bool)
    want[3:10, 4:13] = True
    np.testing.assert_array_equal(m, want)
    for ang in (0, 30, 77, 90, 135, 200, 359):
        m = ref.fill_convex_poly_mask(90, 110, ref.ellipse2poly((52, 44), (25, 4), ang))
        yy, xx = np.mgrid[0:90, 0:110]
        co, si = np.cos(np.deg2rad(ang)), np.sin(np.deg2rad(ang))
        u = (xx - 52) * co + (yy - 44) * si
        v = (yy - 44) * co - (xx - 52) * si
        assert np.all(m[(u / 24.0) ** 2 + (v / 3.0) ** 2 <= 1.0]), ang
        assert not np.any(m[(u / 26.5) ** 2 + (v / 5.5) ** 2 > 1.0]), ang
    m = ref.fill_convex_poly_mask(20, 20, ref.ellipse2poly((10, 10), (0, 4), 90))
    assert m.sum() == 9 and m[10, 6:15].all()
    # shapes partly or wholly outside the image do not wrap or raise
    m = ref.fill_convex_poly_mask(20, 20, ref.ellipse2poly((-3, 5), (8, 4), 20))
    assert m[:, :8].any() and not m[:, 12:].any()
    assert not ref.fill_convex_poly_mask(20, 20, ref.ellipse2poly((100, 100), (8, 4), 20)).any()
