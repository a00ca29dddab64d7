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
