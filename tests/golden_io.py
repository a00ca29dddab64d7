"""Load tests/golden/*.npz (written by tools/make_golden.py from the reference's
own utils/postprocess.py) back into the reference's Python structures."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')

SCENES = ['scene368_p5', 'scene368_p3', 'scene368_p8', 'scene_184x248_p2', 'scene368_p0',
          'scene656_p30']
PEAK_CASES = ['plateau', 'tiny_9x11', 'f32_48x40', 'neg_40x40']
CONN_CASES = [0, 1, 2, 3]


def load(name):
    return np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False)


def peaks_from(rows, part):
    """(n,4) [c0, c1, score, id] + part index -> list of 18 lists of tuples."""
    out = [[] for _ in range(18)]
    for r, p in zip(rows, part):
        out[int(p)].append((np.int64(r[0]), np.int64(r[1]), np.float64(r[2]), int(r[3])))
    return out


def conns_from(rows, limb, is_list=None):
    out = []
    for k in range(19):
        sel = rows[limb == k]
        if is_list is not None and bool(is_list[k]):
            out.append([])
        else:
            out.append(np.asarray(sel, np.float64).reshape(-1, 5))
    return out


def scene_inputs(g):
    heat = (g['heat_q'].astype(np.float32) / np.float32(4096.0)).astype(np.float32)
    paf = (g['paf_q'].astype(np.float32) / np.float32(4096.0)).astype(np.float32)
    return heat, paf


def assert_peaks_equal(got, want, score_rtol=1e-5, exact_score=False):
    assert len(got) == len(want) == 18
    for p in range(18):
        assert len(got[p]) == len(want[p]), 'part %d: %d vs %d peaks' % (p, len(got[p]), len(want[p]))
        for a, b in zip(got[p], want[p]):
            assert int(a[0]) == int(b[0]) and int(a[1]) == int(b[1]), (p, a, b)
            assert int(a[3]) == int(b[3])
            if exact_score:
                assert float(a[2]) == float(b[2])
            else:
                assert abs(float(a[2]) - float(b[2])) <= score_rtol * abs(float(b[2])) + 1e-300


def assert_conns_equal(got, want, score_rtol=1e-5):
    assert len(got) == len(want) == 19
    for k in range(19):
        g, w = got[k], want[k]
        assert isinstance(g, list) == isinstance(w, list), 'limb %d container type' % k
        assert len(g) == len(w), 'limb %d: %d vs %d connections' % (k, len(g), len(w))
        if isinstance(w, list):
            continue
        g = np.asarray(g, np.float64).reshape(-1, 5)
        w = np.asarray(w, np.float64).reshape(-1, 5)
        assert g.shape == w.shape
        np.testing.assert_array_equal(g[:, :2], w[:, :2], err_msg='limb %d ids' % k)
        np.testing.assert_allclose(g[:, 2], w[:, 2], rtol=score_rtol, atol=0)
        np.testing.assert_allclose(g[:, 3:], w[:, 3:], rtol=score_rtol, atol=0)


def assert_persons_equal(got, want, score_rtol=1e-5):
    got = np.asarray(got, np.float64).reshape(-1, 20)
    want = np.asarray(want, np.float64).reshape(-1, 20)
    assert got.shape == want.shape, (got.shape, want.shape)
    np.testing.assert_array_equal(got[:, :18], want[:, :18])
    np.testing.assert_array_equal(got[:, 19], want[:, 19])
    np.testing.assert_allclose(got[:, 18], want[:, 18], rtol=score_rtol, atol=0)
